/* placeholder, filled below */
#include "mso.h"
