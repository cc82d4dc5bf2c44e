// test_frontend.cpp -- the reference's own tests, written against include/recfilter.hpp.
//
// Each case follows one file of /root/reference/tests: build the filter with the RecFilter
// front-end, tile it, realize it on the GPU, compute the expected image with plain raster loops in
// this file (the form the reference's tests use) and compare.  Unlike the reference's tests, which
// only print the error, this exits non-zero when the maximum error exceeds the 1e-4 parity bar.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "recfilter.hpp"

static int failures = 0;

template <typename T>
static T *upload(const std::vector<T> &h) {
    T *d = nullptr;
    if (hipMalloc(&d, h.size() * sizeof(T)) != hipSuccess) { std::fprintf(stderr, "hipMalloc failed\n"); std::exit(2); }
    if (hipMemcpy(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice) != hipSuccess) std::exit(2);
    return d;
}

static std::vector<float> random_image(size_t n, unsigned seed) {
    std::vector<float> v(n);
    unsigned long long s = 0x9E3779B97F4A7C15ull * (seed + 1);
    for (auto &x : v) {       // SplitMix64 -> [0,1)
        s += 0x9E3779B97F4A7C15ull;
        unsigned long long z = s;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        x = (float)(z >> 40) / 16777216.0f;
    }
    return v;
}

// max |ref-out| / max(|ref|, 1% of peak): the parity metric of tests/ref_cases.py
static double rel_err(const std::vector<float> &ref, const std::vector<float> &out) {
    double peak = 0;
    for (float r : ref) peak = std::fmax(peak, std::fabs(r));
    double worst = 0;
    for (size_t i = 0; i < ref.size(); i++)
        worst = std::fmax(worst, std::fabs((double)ref[i] - out[i]) / std::fmax(std::fabs(ref[i]), 1e-2 * peak));
    return worst;
}

static void report(const char *name, double err, double tol = 1e-4) {
    std::printf("%-34s max rel err %.3e %s\n", name, err, err < tol ? "ok" : "FAILED");
    if (!(err < tol)) failures++;
}

// zero-border scan loops in the style of tests/test_generic_xy.cpp:62-110
static void loop_scan(std::vector<float> &ref, int w, int h, int c, int dim, bool causal, const std::vector<float> &W) {
    const int ext[3] = {w, h, c};
    const int n = ext[dim];
    const long stride[3] = {1, w, (long)w * h};
    for (int z = 0; z < c; z++) for (int y = 0; y < h; y++) for (int x = 0; x < w; x++) {
        int idx[3] = {x, y, z};
        if (idx[dim] != 0) continue;                     // one pass per line
        long base = x * stride[0] + y * stride[1] + z * stride[2];
        for (int r = 0; r < n; r++) {
            int i = causal ? r : n - 1 - r;
            float acc = 0.0f;
            for (size_t j = 1; j < W.size(); j++)
                if (r > (int)j - 1) acc += W[j] * ref[base + (causal ? i - (long)j : i + (long)j) * stride[dim]];
            ref[base + i * stride[dim]] = W[0] * ref[base + i * stride[dim]] + acc;
        }
    }
}

int main() {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { std::fprintf(stderr, "no GPU\n"); return 2; }

    {   // tests/test_trivial.cpp: 20x20, tile 4, summed-area table of ones = (x+1)(y+1)
        const int width = 20, height = 20, tile = 4;
        std::vector<float> image(width * height, 1.0f);
        float *d = upload(image);
        RecFilterDim x("x", width), y("y", height);
        RecFilter filter;
        filter(x, y) = RecFilterImage(d);
        filter.add_filter(+x, {1.0f, 1.0f});
        filter.add_filter(+y, {1.0f, 1.0f});
        filter.split(x, tile, y, tile);
        std::vector<float> out = filter.realize().to_host<float>();
        std::vector<float> ref(width * height);
        for (int j = 0; j < height; j++) for (int i = 0; i < width; i++) ref[j * width + i] = float((i + 1) * (j + 1));
        report("test_trivial", rel_err(ref, out), 1e-7);
        (void)hipFree(d);
    }
    {   // tests/test_generic_xy.cpp: 16x16, tile 4, seven order-2 scans
        const int width = 16, height = 16, tile = 4;
        std::vector<float> image = random_image(width * height, 1);
        float *d = upload(image);
        const float W[7][2] = {{0.5f, 0.25f}, {0.5f, 0.125f}, {0.5f, 0.0625f}, {0.5f, 0.125f}, {0.5f, 0.25f}, {0.5f, 0.0625f}, {0.5f, 0.125f}};
        RecFilterDim x("x", width), y("y", height);
        RecFilter filter;
        filter(x, y) = RecFilterImage(d);
        filter.add_filter(+x, {1.0f, W[0][0], W[0][1]});
        filter.add_filter(-x, {1.0f, W[1][0], W[1][1]});
        filter.add_filter(+x, {1.0f, W[2][0], W[2][1]});
        filter.add_filter(-x, {1.0f, W[3][0], W[3][1]});
        filter.add_filter(+y, {1.0f, W[4][0], W[4][1]});
        filter.add_filter(-y, {1.0f, W[5][0], W[5][1]});
        filter.add_filter(-y, {1.0f, W[6][0], W[6][1]});
        filter.split(x, tile, y, tile);
        std::vector<float> out = filter.realize().to_host<float>();
        std::vector<float> ref = image;
        const int dims[7] = {0, 0, 0, 0, 1, 1, 1};
        const bool causal[7] = {true, false, true, false, true, false, false};
        for (int s = 0; s < 7; s++) loop_scan(ref, width, height, 1, dims[s], causal[s], {1.0f, W[s][0], W[s][1]});
        report("test_generic_xy", rel_err(ref, out));
        (void)hipFree(d);
    }
    {   // tests/test_generic_xyz.cpp: 16^3, tile 4, six order-2 scans
        const int n = 16, tile = 4;
        std::vector<float> image = random_image((size_t)n * n * n, 2);
        float *d = upload(image);
        const float W[6][2] = {{0.5f, 0.25f}, {0.5f, 0.125f}, {0.5f, 0.0625f}, {0.5f, 0.125f}, {0.5f, 0.25f}, {0.5f, 0.0625f}};
        RecFilterDim x("x", n), y("y", n), z("z", n);
        RecFilter filter;
        filter(x, y, z) = RecFilterImage(d);
        filter.add_filter(+x, {1.0f, W[0][0], W[0][1]});
        filter.add_filter(-x, {1.0f, W[1][0], W[1][1]});
        filter.add_filter(+y, {1.0f, W[2][0], W[2][1]});
        filter.add_filter(-y, {1.0f, W[3][0], W[3][1]});
        filter.add_filter(+z, {1.0f, W[4][0], W[4][1]});
        filter.add_filter(-z, {1.0f, W[5][0], W[5][1]});
        filter.split(x, tile, y, tile, z, tile);
        std::vector<float> out = filter.realize().to_host<float>();
        std::vector<float> ref = image;
        for (int s = 0; s < 6; s++) loop_scan(ref, n, n, n, s / 2, s % 2 == 0, {1.0f, W[s][0], W[s][1]});
        report("test_generic_xyz", rel_err(ref, out));
        (void)hipFree(d);
    }
    {   // tests/test_type_invariance.cpp: int16, coefficients cast to the pixel type, bit-exact
        const int width = 20, height = 20, tile = 4;
        std::vector<int16_t> image(width * height);
        for (size_t i = 0; i < image.size(); i++) image[i] = (int16_t)((i * 37 + 11) % 19);
        int16_t *d = upload(image);
        RecFilterDim x("x", width), y("y", height);
        RecFilter filter;
        filter(x, y) = RecFilterImage(d);
        filter.add_filter(+x, {1.0f, 1.0f, -1.0f});
        filter.add_filter(+y, {1.0f, 1.0f, -1.0f});
        filter.split(x, tile, y, tile);
        std::vector<int16_t> out = filter.realize().to_host<int16_t>();
        std::vector<int16_t> ref = image;
        for (int j = 0; j < height; j++) for (int i = 0; i < width; i++)
            ref[j * width + i] = (int16_t)(ref[j * width + i] + (i > 0 ? ref[j * width + i - 1] : 0) - (i > 1 ? ref[j * width + i - 2] : 0));
        for (int j = 0; j < height; j++) for (int i = 0; i < width; i++)
            ref[j * width + i] = (int16_t)(ref[j * width + i] + (j > 0 ? ref[(j - 1) * width + i] : 0) - (j > 1 ? ref[(j - 2) * width + i] : 0));
        int bad = 0;
        for (size_t i = 0; i < ref.size(); i++) bad += ref[i] != out[i];
        std::printf("%-34s %d mismatching pixels %s\n", "test_type_invariance (int16)", bad, bad ? "FAILED" : "ok");
        failures += bad != 0;
        (void)hipFree(d);
    }
    {   // tests/test_overlap_filter_order.cpp: cascade of two filters == one higher-order filter
        const int width = 12, height = 12;
        std::vector<float> image = random_image(width * height, 3);
        float *d = upload(image);
        RecFilterDim x("x", width), y("y", height);
        RecFilter f1("R1");
        f1(x, y) = RecFilterImage(d);
        f1.add_filter(+x, {1.0f, 2.0f, -1.0f});
        f1.add_filter(+y, {1.0f, 1.0f});
        RecFilter f2("R2");
        f2(x, y) = f1;
        f2.add_filter(+x, {1.0f, 1.0f});
        f2.add_filter(+y, {1.0f, 2.0f, -1.0f});
        RecFilter f3 = f2.overlap_to_higher_order_filter(f1, "O");
        std::vector<float> ref = f2.realize().to_host<float>();
        std::vector<float> out = f3.realize().to_host<float>();
        report("test_overlap_filter_order", rel_err(ref, out));
        CheckResult<float> cr(ref, out);          // the reference's own summary (percent)
        std::printf("%-34s CheckResult: max %.2e %%, mean %.2e %%\n", "", cr.max_diff, cr.mean_diff);
        failures += !(cr.mean_diff < 1e-2f);
        (void)hipFree(d);
    }
    {   // apps/gaussian/gaussian_filter_1xy_2xy.cpp: 8 scans cascaded {0-3},{4-7}, clamped, 1024^2 -> fused path
        const int width = 1024;
        std::vector<float> image = random_image((size_t)width * width, 4);
        float *d = upload(image);
        RecFilterDim x("x", width), y("y", width);
        std::vector<float> W1 = gaussian_weights(5.0f, 1), W2 = gaussian_weights(5.0f, 2);
        RecFilter F("Gaussian_1xy_2xy");
        F.set_clamped_image_border();
        F(x, y) = RecFilterImage(d);
        F.add_filter(+x, W1); F.add_filter(-x, W1); F.add_filter(+y, W1); F.add_filter(-y, W1);
        F.add_filter(+x, W2); F.add_filter(-x, W2); F.add_filter(+y, W2); F.add_filter(-y, W2);
        std::vector<RecFilter> fc = F.cascade({0, 1, 2, 3}, {4, 5, 6, 7});
        for (auto &f : fc) { f.split_all_dimensions(32); RecFilter::set_max_threads_per_cuda_warp(128); f.gpu_auto_schedule(); }
        std::vector<float> out = fc.back().realize().to_host<float>();
        // untiled version of the same eight scans is the reference result
        RecFilter U("untiled");
        U.set_clamped_image_border();
        U(x, y) = RecFilterImage(d);
        for (const auto &W : {W1, W2}) { U.add_filter(+x, W); U.add_filter(-x, W); U.add_filter(+y, W); U.add_filter(-y, W); }
        std::vector<float> ref = U.realize().to_host<float>();
        report("gaussian_1xy_2xy (tiled vs untiled)", rel_err(ref, out));
        float ms = fc.back().profile(5);
        std::printf("%-34s %.3f ms per realize (both stages, 1024^2)\n", "profile()", ms);
        failures += !(ms > 0.0f);
        (void)hipFree(d);
    }
    {   // apps/audio/audio_filter_high_order.cpp:38-73: one causal scan of order 1, 3, .. 29 ("dummy coeff": 0.01) over a 1-D
        // signal, tiled and non-tiled; orders above 8 run in their direct form on the matrix cores (RF_PATH_TILED_MATRIX)
        const int width = 1 << 18, tile_width = 32;
        std::vector<float> image = random_image((size_t)width, 7);
        float *d = upload(image);
        double worst = 0.0;
        int high_path = -1;
        for (int order = 1; order < 30; order += 2) {
            std::vector<float> coeffs(order + 1, 0.01f);
            coeffs[0] = 1.0f;
            RecFilterDim x("x", width);
            RecFilter F("R_tiled");
            F(x) = RecFilterImage(d);
            F.add_filter(+x, coeffs);
            F.split(x, tile_width);
            std::vector<float> out = F.realize().to_host<float>();
            std::vector<float> ref = image;
            loop_scan(ref, width, 1, 1, 0, true, coeffs);
            worst = std::max(worst, rel_err(ref, out));
            if (order == 29) {
                const std::string syn = F.print_synopsis();
                const size_t at = syn.find("plan: path ");
                if (at != std::string::npos) high_path = std::atoi(syn.c_str() + at + 11);
            }
        }
        report("audio_high_order (orders 1..29)", worst);
        std::printf("%-34s order 29 runs on path %d %s\n", "", high_path, high_path == RF_PATH_TILED_MATRIX ? "ok" : "FAILED");
        failures += high_path != RF_PATH_TILED_MATRIX;
        (void)hipFree(d);
    }
    {   // apps/usm/unsharp_mask_optimized.cpp: USM = (1+w)*I - w*Blur(I), Blur computed at USM's tiles; the input is
        // defined as image/255 like demo/demo_gaussian_filter.cpp:51-53.  Reference result: untiled blur + host loop.
        const int width = 512, height = 256;
        const float weight = 1.0f;
        std::vector<float> image = random_image((size_t)width * height, 5);
        std::vector<uint8_t> bytes(image.size());
        for (size_t i = 0; i < image.size(); i++) { bytes[i] = (uint8_t)(image[i] * 255.0f); image[i] = (float)bytes[i]; }
        uint8_t *d = upload(bytes);          // the filters read the uint8 image directly
        RecFilterDim x("x", width), y("y", height);
        std::vector<float> W3 = gaussian_weights(5.0f, 3);
        RecFilter B("Blur"), U("Blur_untiled");
        for (RecFilter *f : {&B, &U}) {
            f->set_clamped_image_border();
            (*f)(x, y) = RecFilterImage(d) / 255.0f;
            f->add_filter(+x, W3); f->add_filter(-x, W3); f->add_filter(+y, W3); f->add_filter(-y, W3);
        }
        B.split_all_dimensions(32);
        B.compute_at(RecFilterPointwise{-weight, 1.0f + weight, 0.0f});
        std::vector<float> out = B.realize().to_host<float>();
        std::vector<float> ref = U.realize().to_host<float>();
        // the mask is a difference of two O(1) terms: the 1e-4 bar is relative to the terms it combines
        double worst = 0, peak = 0;
        std::vector<double> scale(ref.size());
        for (size_t i = 0; i < ref.size(); i++) {
            scale[i] = (1.0 + weight) * std::fabs(image[i] / 255.0f) + weight * std::fabs(ref[i]);
            peak = std::fmax(peak, scale[i]);
            ref[i] = (1.0f + weight) * (image[i] / 255.0f) - weight * ref[i];
        }
        for (size_t i = 0; i < ref.size(); i++)
            worst = std::fmax(worst, std::fabs((double)ref[i] - out[i]) / std::fmax(scale[i], 1e-2 * peak));
        report("unsharp_mask (compute_at, in/255)", worst);
        (void)hipFree(d);
    }
    {   // misuse behaves like the reference's assert(false) sites, as exceptions
        int caught = 0;
        RecFilterDim x("x", 16), y("y", 16);
        std::vector<float> image(256, 1.0f);
        float *d = upload(image);
        auto expect_throw = [&](const std::function<void()> &fn) { try { fn(); } catch (const RecFilterError &) { caught++; } };
        RecFilter f;
        expect_throw([&] { f.add_filter(+x, {1.0f, 0.5f}); });                 // lib/recfilter.cpp:268-272
        f(x, y) = RecFilterImage(d);
        expect_throw([&] { f(x, y) = RecFilterImage(d); });                    // :205-208
        expect_throw([&] { f.set_clamped_image_border(); });                    // :252-256
        expect_throw([&] { f.add_filter(+x, {1.0f}); });                        // :274-278
        expect_throw([&] { f.add_filter(+RecFilterDim("w", 4), {1.0f, 0.5f}); }); // :296-300
        f.add_filter(+x, {1.0f, 0.5f});
        f.add_filter(-x, {1.0f, 0.5f});
        expect_throw([&] { f.split(y, 4); });                                   // lib/split.cpp:1879-1883
        expect_throw([&] { f.cascade({1}, {0}); });                             // lib/reorder.cpp:70-75
        f.split(x, 4);
        expect_throw([&] { f.split(x, 4); });                                   // lib/split.cpp:1851-1854
        expect_throw([&] { f.cascade({0}, {1}); });                             // lib/reorder.cpp:29-33
        expect_throw([&] { f.full_schedule(); });                               // lib/recfilter.cpp:397-401
        f.intra_schedule(1).compute_locally().unroll(0).gpu_threads(0, 1);
        std::printf("%-34s %d/10 misuse cases rejected %s\n", "front-end misuse", caught, caught == 10 ? "ok" : "FAILED");
        failures += caught != 10;
        (void)hipFree(d);
    }
    {   // two filters enqueued on two streams (set_stream / enqueue): each keeps its own result
        const int width = 512, height = 256;
        const std::vector<float> W = gaussian_weights(5.0f, 2);
        hipStream_t st[2];
        std::vector<float> images[2] = {random_image((size_t)width * height, 31), random_image((size_t)width * height, 32)};
        float *d[2];
        RecFilterDim x("x", width), y("y", height);
        std::vector<RecFilter> filters;
        for (int i = 0; i < 2; i++) {
            if (hipStreamCreate(&st[i]) != hipSuccess) { std::fprintf(stderr, "hipStreamCreate failed\n"); return 2; }
            d[i] = upload(images[i]);
            RecFilter f;
            f.set_clamped_image_border();
            f(x, y) = RecFilterImage(d[i]);
            f.add_filter(+x, W); f.add_filter(-x, W); f.add_filter(+y, W); f.add_filter(-y, W);
            f.split_all_dimensions(32);
            f.set_stream(st[i]);
            filters.push_back(f);
        }
        RecFilterRealization r[2];
        for (int rep = 0; rep < 3; rep++)
            for (int i = 0; i < 2; i++) r[i] = filters[i].enqueue();
        for (int i = 0; i < 2; i++) {
            if (hipStreamSynchronize(st[i]) != hipSuccess) { std::fprintf(stderr, "hipStreamSynchronize failed\n"); return 2; }
            std::vector<float> ref = images[i];
            // the clamped loops of the reference's apps: taps before the first sample read the sample itself
            for (int dim = 0; dim < 2; dim++)
                for (int causal = 1; causal >= 0; causal--) {
                    const int n = dim == 0 ? width : height, lines = dim == 0 ? height : width;
                    for (int ln = 0; ln < lines; ln++)
                        for (int p = 0; p < n; p++) {
                            const int i0 = causal ? p : n - 1 - p;
                            auto at = [&](int q) -> float & { return dim == 0 ? ref[(size_t)ln * width + q] : ref[(size_t)q * width + ln]; };
                            float acc = W[0] * at(i0);
                            for (int j = 1; j <= 2; j++) {
                                int q = causal ? i0 - j : i0 + j;
                                q = q < 0 ? 0 : (q > n - 1 ? n - 1 : q);
                                acc += W[j] * at(q);
                            }
                            at(i0) = acc;
                        }
                }
            report(i == 0 ? "enqueue on stream 0" : "enqueue on stream 1", rel_err(ref, r[i].to_host<float>()));
            (void)hipFree(d[i]);
        }
        filters.clear();
        for (int i = 0; i < 2; i++) (void)hipStreamDestroy(st[i]);
    }
    {   // the difference Funcs the apps put behind a summed-area table (apps/box/box_filter.h:36-39, apps/DoG/diff_gauss.cpp:176-197):
        // box_difference() and the generic tap_filter() on the realization of a summed-area table, against clamped loops
        const int width = 96, height = 64, B = 3;
        std::vector<float> image = random_image((size_t)width * height, 21);
        float *d = upload(image);
        RecFilterDim x("x", width), y("y", height);
        RecFilter sat;
        sat(x, y) = RecFilterImage(d);
        sat.add_filter(+x, {1.0f, 1.0f});
        sat.add_filter(+y, {1.0f, 1.0f});
        sat.split_all_dimensions(32);
        RecFilterRealization table = sat.realize();
        std::vector<float> t = table.to_host<float>();
        float *out1 = nullptr, *out2 = nullptr;
        if (hipMalloc(&out1, t.size() * sizeof(float)) != hipSuccess || hipMalloc(&out2, t.size() * sizeof(float)) != hipSuccess) return 2;
        box_difference(table, 0, out1, B, {1, 1});
        const float s = 1.0f / float((2 * B + 1) * (2 * B + 1));
        std::vector<rf_tap> taps = {{0, {B, B, 0}, s}, {0, {B, -B - 1, 0}, -s}, {0, {-B - 1, -B - 1, 0}, s}, {0, {-B - 1, B, 0}, -s}};
        tap_filter({table.planes[0]}, out2, table.extent, RF_F32, taps);
        if (hipDeviceSynchronize() != hipSuccess) return 2;
        std::vector<float> h1(t.size()), h2(t.size()), ref(t.size());
        (void)hipMemcpy(h1.data(), out1, t.size() * sizeof(float), hipMemcpyDeviceToHost);
        (void)hipMemcpy(h2.data(), out2, t.size() * sizeof(float), hipMemcpyDeviceToHost);
        auto at = [&](int xx, int yy) { xx = xx < 0 ? 0 : (xx > width - 1 ? width - 1 : xx); yy = yy < 0 ? 0 : (yy > height - 1 ? height - 1 : yy);
                                        return (double)t[(size_t)yy * width + xx]; };
        for (int yy = 0; yy < height; yy++) for (int xx = 0; xx < width; xx++)
            ref[(size_t)yy * width + xx] = (float)((at(xx + B, yy + B) - at(xx + B, yy - B - 1) + at(xx - B - 1, yy - B - 1) - at(xx - B - 1, yy + B)) * s);
        report("tap_filter (diff_op_xy)", rel_err(ref, h2), 1e-4);
        // box_difference nests the clamps per dimension (box_filter.h: fA inside diff): equal to the four-tap form away
        // from the borders, so compare the interior
        double worst = 0;
        for (int yy = 2 * B + 2; yy < height - 2 * B - 2; yy++) for (int xx = 2 * B + 2; xx < width - 2 * B - 2; xx++)
            worst = std::fmax(worst, std::fabs((double)h1[(size_t)yy * width + xx] - ref[(size_t)yy * width + xx]));
        report("box_difference (interior)", worst, 1e-3);
        (void)hipFree(out1); (void)hipFree(out2); (void)hipFree(d);
    }
    {   // sharded realization from C++ (no counterpart in the reference): three ranks as three threads on this one device,
        // row slabs of different heights, the all-gather emulated with device copies between two thread barriers.  With
        // RCCL the callback is ncclAllGather(send, gathered, bytes, ncclChar, comm, (hipStream_t)stream).
        const int width = 512, world = 3;
        const std::vector<int64_t> extents = {128, 64, 192};
        const int height = 128 + 64 + 192;
        std::vector<float> image = random_image((size_t)width * height, 11);
        const std::vector<float> W = {0.30f, 0.90f, -0.25f};
        struct Barrier {
            std::mutex m; std::condition_variable cv; int count = 0, generation = 0, n;
            explicit Barrier(int n_) : n(n_) {}
            void wait() {
                std::unique_lock<std::mutex> lk(m);
                const int g = generation;
                if (++count == n) { count = 0; generation++; cv.notify_all(); }
                else cv.wait(lk, [&] { return g != generation; });
            }
        } barrier(world);
        std::vector<const void *> sends(world, nullptr);
        std::vector<std::vector<float>> outs(world);
        std::vector<std::string> errors(world);
        std::vector<int> exchanges(world, -1);
        auto rank_main = [&](int rank) {
            try {
                hipStream_t st;
                if (hipStreamCreate(&st) != hipSuccess) throw RecFilterError("hipStreamCreate failed");
                int64_t lo = 0;
                for (int r = 0; r < rank; r++) lo += extents[r];
                std::vector<float> slab(image.begin() + lo * width, image.begin() + (lo + extents[rank]) * width);
                float *d = upload(slab);
                RecFilterDim x("x", width), y("y", (int)extents[rank]);
                RecFilter F;
                F(x, y) = RecFilterImage(d);
                F.add_filter(+x, W); F.add_filter(-x, W); F.add_filter(+y, W); F.add_filter(-y, W);
                F.split_all_dimensions(32);
                F.shard(rank, world, extents);
                F.set_stream(st);
                int n_calls = 0;
                RecFilter::AllGather gather = [&](const void *send, void *gathered, size_t bytes, void *stream) {
                    n_calls++;
                    sends[rank] = send;
                    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) throw RecFilterError("sync failed");
                    barrier.wait();                         // every rank's exit carries are written
                    for (int r = 0; r < world; r++)
                        if (hipMemcpyAsync((char *)gathered + (size_t)r * bytes, sends[r], bytes, hipMemcpyDeviceToDevice,
                                           (hipStream_t)stream) != hipSuccess) throw RecFilterError("copy failed");
                    if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) throw RecFilterError("sync failed");
                    barrier.wait();                         // ... and read by everybody
                };
                outs[rank] = F.realize_sharded(gather).to_host<float>();
                exchanges[rank] = n_calls;
                (void)hipFree(d);
                (void)hipStreamDestroy(st);
            } catch (const std::exception &e) { errors[rank] = e.what(); }
        };
        std::vector<std::thread> threads;
        for (int r = 0; r < world; r++) threads.emplace_back(rank_main, r);
        for (auto &t : threads) t.join();
        bool ok = true;
        for (int r = 0; r < world; r++)
            if (!errors[r].empty()) { std::fprintf(stderr, "rank %d: %s\n", r, errors[r].c_str()); ok = false; }
        if (!ok) { failures++; std::printf("%-34s FAILED (exception)\n", "sharded realization, 3 ranks"); }
        else {
            std::vector<float> ref = image, got;
            loop_scan(ref, width, height, 1, 0, true, W); loop_scan(ref, width, height, 1, 0, false, W);
            loop_scan(ref, width, height, 1, 1, true, W); loop_scan(ref, width, height, 1, 1, false, W);
            for (int r = 0; r < world; r++) got.insert(got.end(), outs[r].begin(), outs[r].end());
            report("sharded realization, 3 ranks", rel_err(ref, got));
            if (exchanges[0] != 1) { failures++; std::printf("expected ONE all-gather per execution, saw %d\n", exchanges[0]); }
        }
    }
    {   // a z-sharded volume from C++: two ranks as two threads; the plan exchanges the z carries of the RAW input and has
        // exchange-independent work (the x/y stage), so the front-end hands the collective a SIDE stream and runs
        // rf_plan_interior beside it.  Then ONE slab driven through the same calls (RF_PLAN_FORCE_EXCHANGE).
        const int nx = 256, ny = 48, world = 2;
        const std::vector<int64_t> extents = {64, 32};
        const int nz = 96;
        std::vector<float> image = random_image((size_t)nx * ny * nz, 17);
        const std::vector<float> W = {0.40f, 0.70f, -0.20f};
        struct Barrier {
            std::mutex m; std::condition_variable cv; int count = 0, generation = 0, n;
            explicit Barrier(int n_) : n(n_) {}
            void wait() {
                std::unique_lock<std::mutex> lk(m);
                const int g = generation;
                if (++count == n) { count = 0; generation++; cv.notify_all(); }
                else cv.wait(lk, [&] { return g != generation; });
            }
        };
        std::vector<float> ref = image;
        for (int dim = 0; dim < 3; dim++) { loop_scan(ref, nx, ny, nz, dim, true, W); loop_scan(ref, nx, ny, nz, dim, false, W); }
        for (int mode = 0; mode < 2; mode++) {          // 0: two slabs; 1: one slab with the exchange structure forced
            const int ranks = mode == 0 ? world : 1;
            Barrier barrier(ranks);
            std::vector<const void *> sends(ranks, nullptr);
            std::vector<std::vector<float>> outs(ranks);
            std::vector<std::string> errors(ranks);
            std::vector<int> side_stream_calls(ranks, 0);
            auto rank_main = [&](int rank) {
                try {
                    hipStream_t st;
                    if (hipStreamCreate(&st) != hipSuccess) throw RecFilterError("hipStreamCreate failed");
                    const int64_t lo = mode == 0 ? (rank == 0 ? 0 : extents[0]) : 0;
                    const int64_t mine = mode == 0 ? extents[rank] : nz;
                    std::vector<float> slab(image.begin() + lo * nx * ny, image.begin() + (lo + mine) * nx * ny);
                    float *d = upload(slab);
                    RecFilterDim x("x", nx), y("y", ny), z("z", (int)mine);
                    RecFilter F;
                    F(x, y, z) = RecFilterImage(d);
                    F.add_filter(+x, W); F.add_filter(-x, W); F.add_filter(+y, W); F.add_filter(-y, W);
                    F.add_filter(+z, W); F.add_filter(-z, W);
                    F.split(x, 32, y, 16, z, 32);
                    if (mode == 0) F.shard(rank, world, extents);
                    else F.plan_options(RF_PLAN_FORCE_EXCHANGE | RF_PLAN_TILED_ONLY);
                    F.set_stream(st);
                    RecFilter::AllGather gather = [&](const void *send, void *gathered, size_t bytes, void *stream) {
                        if ((hipStream_t)stream != st) side_stream_calls[rank]++;
                        sends[rank] = send;
                        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) throw RecFilterError("sync failed");
                        barrier.wait();
                        for (int r = 0; r < ranks; r++)
                            if (hipMemcpyAsync((char *)gathered + (size_t)r * bytes, sends[r], bytes, hipMemcpyDeviceToDevice,
                                               (hipStream_t)stream) != hipSuccess) throw RecFilterError("copy failed");
                        if (hipStreamSynchronize((hipStream_t)stream) != hipSuccess) throw RecFilterError("sync failed");
                        barrier.wait();
                    };
                    outs[rank] = F.realize_sharded(gather).to_host<float>();
                    (void)hipFree(d);
                    (void)hipStreamDestroy(st);
                } catch (const std::exception &e) { errors[rank] = e.what(); }
            };
            std::vector<std::thread> threads;
            for (int r = 0; r < ranks; r++) threads.emplace_back(rank_main, r);
            for (auto &t : threads) t.join();
            const char *label = mode == 0 ? "z-sharded volume, 2 ranks" : "one slab, exchange forced";
            bool ok = true;
            for (int r = 0; r < ranks; r++)
                if (!errors[r].empty()) { std::fprintf(stderr, "rank %d: %s\n", r, errors[r].c_str()); ok = false; }
            if (!ok) { failures++; std::printf("%-34s FAILED (exception)\n", label); continue; }
            std::vector<float> got;
            for (int r = 0; r < ranks; r++) got.insert(got.end(), outs[r].begin(), outs[r].end());
            report(label, rel_err(ref, got));
            if (side_stream_calls[0] != 1) { failures++; std::printf("%s: expected the collective on a side stream (interior beside it), saw %d\n", label, side_stream_calls[0]); }
        }
    }
    std::printf("%s\n", failures ? "SOME TESTS FAILED" : "all front-end tests passed");
    return failures ? 1 : 0;
}
