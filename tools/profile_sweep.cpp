// profile_sweep.cpp -- the reference's benchmark methodology from C++ (scripts/profile_app.sh:6-19: widths 64 .. 4096
// step 64, tile 32; lib/recfilter.cpp:991-1016: profile(iterations)), on include/recfilter.hpp: what a C++ caller pays
// per image, without the Python layer the other tools go through.
//   tools/profile_sweep [app] [iterations] [max width]      app = gaussian_3xy | summed_table | bicubic
// prints  width<TAB>ms<TAB>MiP/s  rows (lib/timing.cpp:3-5: MiP/s = w*w*1000 / (ms * 2^20))
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "recfilter.hpp"

int main(int argc, char **argv) {
    const char *app = argc > 1 ? argv[1] : "gaussian_3xy";
    const int iterations = argc > 2 ? std::atoi(argv[2]) : 200;
    const int max_width = argc > 3 ? std::atoi(argv[3]) : 4096;
    float *in = nullptr;
    if (hipMalloc(&in, (size_t)max_width * max_width * sizeof(float)) != hipSuccess) return 2;
    std::vector<float> h((size_t)max_width * max_width);
    for (size_t i = 0; i < h.size(); i++) h[i] = (float)((i * 2654435761u) >> 8 & 0xffff) / 65536.0f;
    if (hipMemcpy(in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return 2;
    for (int w = 64; w <= max_width; w += 64) {
        RecFilterDim x("x", w), y("y", w);
        RecFilter F("F");
        std::vector<float> W;
        if (!std::strcmp(app, "summed_table")) {
            W = {1.0f, 1.0f};
            F(x, y) = RecFilterImage<float>(in);
            F.add_filter(+x, W); F.add_filter(+y, W);
        } else {
            if (!std::strcmp(app, "bicubic")) { const float a = 2.0f - std::sqrt(3.0f); W = {1.0f + a, -a}; }
            else W = gaussian_weights(5.0f, 3);
            F.set_clamped_image_border();
            F(x, y) = RecFilterImage<float>(in);
            F.add_filter(+x, W); F.add_filter(-x, W); F.add_filter(+y, W); F.add_filter(-y, W);
        }
        F.split_all_dimensions(32);                         // the reference's sweep tiles at 32
        const float ms = F.profile(iterations);
        std::printf("%d\t%f\t%.3f\n", w, ms, (double)w * w * 1000.0 / (ms * 1048576.0));
    }
    return 0;
}
