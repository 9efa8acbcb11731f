// CPU sanitizer harness for the host-side geometry of libck_hip.so (ck_host_geom.cpp: convex hull + float32
// rotating calipers, perspective transform, 3x3 inverse).  Built with -fsanitize=address,undefined and fed
// degenerate and random point sets; GPU sanitizers are not available on the pool, so this is where the
// product's host C++ gets its memory / UB check.   tools/sanitize/run.sh builds and runs it.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <random>
#include <vector>

void ck_invert3x3(const double* s, double* d);
void ck_min_area_rect(const int32_t* pts, int n, float* out_wh);
extern "C" int ck_get_perspective_transform(const float* src4, const float* dst4, double* M9);

int main()
{
    std::mt19937 rng(20161001);
    long checks = 0;
    for (int trial = 0; trial < 20000; trial++) {
        const int kind = trial % 8;
        int n = 1 + (int)(rng() % (kind == 7 ? 4000 : 40));
        std::vector<int32_t> p((size_t)n * 2);
        for (int i = 0; i < n; i++) {
            int x = (int)(rng() % 2000), y = (int)(rng() % 1200);
            if (kind == 0) y = 7;                        // all on one horizontal line
            if (kind == 1) x = 13;                       // vertical line
            if (kind == 2) { x = 100; y = 100; }         // one repeated point
            if (kind == 3) y = x;                        // diagonal
            if (kind == 4) { x = (i % 2) * 50; y = (i / 2 % 2) * 30; }   // the 4 corners of a rectangle, repeated
            p[2 * i] = x; p[2 * i + 1] = y;
        }
        float wh[2] = {-1.f, -1.f};
        ck_min_area_rect(p.data(), n, wh);
        if (!(wh[0] >= 0.f && wh[1] >= 0.f) || std::isnan(wh[0]) || std::isnan(wh[1])) {
            fprintf(stderr, "bad rect %g x %g (kind %d, n %d)\n", wh[0], wh[1], kind, n);
            return 1;
        }
        checks++;
    }
    for (int trial = 0; trial < 20000; trial++) {
        float src[8], dst[8] = {0, 0, 380, 0, 380, 380, 0, 380};
        for (int i = 0; i < 8; i++) src[i] = (float)(rng() % 2000) / (trial % 3 == 0 ? 1.f : 7.f);
        if (trial % 5 == 0) { src[2] = src[0]; src[3] = src[1]; }       // two equal corners: degenerate
        double M[9], Mi[9];
        const int rc = ck_get_perspective_transform(src, dst, M);
        if (rc == 0) ck_invert3x3(M, Mi);
        checks++;
    }
    printf("host geometry: %ld calls clean under ASan/UBSan\n", checks);
    return 0;
}
