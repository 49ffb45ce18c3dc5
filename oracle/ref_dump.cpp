// TEST INFRASTRUCTURE ONLY (oracle/_ref): dumps the reference's own ORB pattern table.
//
// This translation unit #includes the reference header WHERE IT LIES
// (/root/reference/openvslam/orb_point_pairs.h:36-47 onward) -- the only hot-path header that
// compiles without OpenCV/Eigen/parent-project headers. The other three small headers
// (match_base.h, trigonometric.h, match_angle_checker.h) #include <opencv2/...> and would need
// stand-in headers, so under the build rules they count as unbuildable here; they are restated
// in oracle/*.c from their source text instead (see DESIGN.md "Oracle pinning").
//
// Output: 1024 signed bytes (the table is integer-valued, range [-13, 13]) on stdout.
#include <cstdio>
#include <cstdint>
#include <cmath>
#include "openvslam/orb_point_pairs.h"

int main() {
    using namespace openvslam::feature;
    static_assert(orb_point_pairs_size == 1024, "pattern size");
    for (unsigned i = 0; i < orb_point_pairs_size; ++i) {
        const float v = orb_point_pairs[i];
        if (v != std::floor(v) || v < -128 || v > 127) { std::fprintf(stderr, "non-integer entry %u\n", i); return 1; }
        const int8_t b = static_cast<int8_t>(v);
        std::fwrite(&b, 1, 1, stdout);
    }
    return 0;
}
