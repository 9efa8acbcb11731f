// ASan / UBSan harness for the host geometry of SfContours.find_stones and the grid-line search
// (camkifu_amd/csrc/ck_stonegeom.cpp + the hull / calipers of ck_host_geom.cpp): random and degenerate polygons through
// the raster (every painted pixel must lie inside the reported box), random masks through the chamfer transform and
// the centre test, random zone tables through find_color, random line bundles through update_grid.
// Built by tools/sanitize/run.sh; no GPU involved.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../../camkifu_amd/csrc/ck_stonegeom.h"

void ck_min_area_rect_box(const int32_t* pts, int n, float* out_wha);
std::vector<int32_t> ck_hull_points(const int32_t* pts, int n);

int main()
{
    std::mt19937 rng(20161001);
    auto uni = [&](int lo, int hi) { return (int)(rng() % (unsigned)(hi - lo + 1)) + lo; };
    long painted = 0, centres = 0, coloured = 0, moved = 0;
    std::vector<uint8_t> bits;
    for (int t = 0; t < 3000; t++) {
        const int n = uni(1, 60), mode = uni(0, 4);
        std::vector<int32_t> p((size_t)2 * n);
        for (int i = 0; i < n; i++) {
            p[2 * i] = mode == 0 ? 5 : uni(0, 378);
            p[2 * i + 1] = mode == 1 ? 9 : (mode == 2 ? p[2 * i] : uni(0, 378));
        }
        const std::vector<int32_t> hull = ck_hull_points(p.data(), n);
        float wha[3];
        ck_min_area_rect_box(hull.data(), (int)(hull.size() / 2), wha);
        if (!(wha[0] >= 0 && wha[1] >= 0 && wha[2] >= -180.f && wha[2] <= 180.f)) { std::puts("box: bad result"); return 1; }
        int bx, by, bw, bh;
        ck_raster_polygon(hull.data(), (int)(hull.size() / 2), &bx, &by, &bw, &bh, bits);
        if (bw <= 0 || bh <= 0 || bits.size() != (size_t)bw * bh) { std::puts("raster: bad box"); return 1; }
        for (size_t i = 0; i + 1 < hull.size(); i += 2)
            if (!bits[(size_t)(hull[i + 1] - by) * bw + hull[i] - bx]) { std::puts("raster: vertex not painted"); return 1; }
        for (uint8_t b : bits) painted += b;
    }
    std::vector<int32_t> dist;
    for (int t = 0; t < 400; t++) {
        const int h = uni(1, 90), w = uni(1, 90);
        std::vector<uint8_t> img((size_t)h * w);
        const int density = uni(0, 100);
        for (auto& v : img) v = uni(0, 99) < density ? 255 : 0;
        ck_chamfer5(img.data(), h, w, dist);
        for (size_t i = 0; i < img.size(); i++)
            if ((img[i] == 0) != (dist[i] == 0)) { std::puts("chamfer: zero set differs"); return 1; }
        const int got = ck_has_stone_center(dist.data(), h, w, 10.0);
        if (got < -1 || got > 1) return 1;
        if ((h < 10 || w < 10) != (got == -1) && !(h == 10 || w == 10)) { std::puts("centres: thin-box rule"); return 1; }
        centres += got > 0;
    }
    for (int t = 0; t < 2000; t++) {
        const int R = uni(1, 19), C = uni(1, 19);
        std::vector<int16_t> zones((size_t)R * C * 4);
        for (size_t i = 0; i < zones.size(); i += 4) {
            zones[i] = (int16_t)uni(0, 1);
            for (int k = 1; k < 4; k++) zones[i + k] = (int16_t)uni(0, 255);
        }
        std::vector<uint8_t> stones(361, 0);
        ck_find_colors(zones.data(), R, C, stones.data(), 19);
        for (uint8_t s : stones) { if (s > 2) return 1; coloured += s != 0; }
    }
    for (int t = 0; t < 20000; t++) {
        const int k = uni(0, 12);
        std::vector<int32_t> lines((size_t)4 * (k ? k : 1));
        for (int i = 0; i < k; i++) {
            do { for (int c = 0; c < 4; c++) lines[4 * i + c] = uni(0, 19); }
            while (lines[4 * i] == lines[4 * i + 2] && lines[4 * i + 1] == lines[4 * i + 3]);
            if (uni(0, 2) == 0) lines[4 * i + 3] = lines[4 * i + 1];           // level
            else if (uni(0, 1) == 0) lines[4 * i + 2] = lines[4 * i];          // upright
            if (lines[4 * i] == lines[4 * i + 2] && lines[4 * i + 1] == lines[4 * i + 3]) lines[4 * i + 2] += 1;
        }
        const int32_t box[4] = { 100, 200, 120, 219 };
        int16_t slot[2] = { 110, 209 };
        ck_update_grid_host(lines.data(), k, box, slot);
        const bool same = slot[0] == 110 && slot[1] == 209, negated = slot[0] == -110 && slot[1] == -209;
        if (!same && !negated) {
            if (!(slot[0] < -100 && slot[0] > -120 && slot[1] < -200 && slot[1] > -219)) { std::puts("update_grid: moved outside its zone"); return 1; }
            moved++;
        }
    }
    std::printf("stonegeom fuzz ok: %ld pixels painted, %ld boxes with a centre, %ld zones coloured, %ld intersections moved\n",
                painted, centres, coloured, moved);
    return 0;
}
