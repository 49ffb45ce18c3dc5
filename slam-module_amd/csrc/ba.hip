// ba.hip -- local bundle adjustment on gfx950 (filled in below in this round).
#include "ms_internal.h"
