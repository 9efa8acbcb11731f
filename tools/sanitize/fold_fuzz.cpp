// ASan / UBSan harness for the ordered halves behind the C-ABI (camkifu_amd/csrc/ck_fold.cpp): random line bundles through
// ck_boardfold_step (including degenerate inputs: no lines, thousands of near-parallel lines, NaN-free extremes), random
// classifier answers / foreground counts through ck_policy_run with every request applied to a toy goban and the run
// resumed, hulls of random and degenerate point sets.  Built by tools/sanitize/run.sh; no GPU involved.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

#include "camkifu_amd.h"

int main()
{
    std::mt19937 rng(20161001);
    auto uni = [&](int lo, int hi) { return (int)(rng() % (unsigned)(hi - lo + 1)) + lo; };
    // ---- hulls
    long hull_pts = 0;
    for (int t = 0; t < 2000; t++) {
        const int n = uni(0, 40);
        std::vector<int32_t> p((size_t)2 * (n ? n : 1)), out((size_t)2 * (n ? n : 1));
        const int mode = uni(0, 3);
        for (int i = 0; i < n; i++) {
            p[2 * i] = mode == 0 ? 7 : uni(-50, 700);                          // all on one vertical line / anywhere
            p[2 * i + 1] = mode == 1 ? p[2 * i] : uni(-50, 500);               // all on the diagonal
        }
        int32_t m = -1;
        if (ck_ordered_hull(p.data(), n, out.data(), &m) != CK_OK || m < 0 || m > (n ? n : 0)) { std::puts("hull: bad result"); return 1; }
        hull_pts += m;
    }
    // ---- board fold
    ck_boardfold* bf = nullptr;
    if (ck_boardfold_create(&bf) != CK_OK) return 1;
    long found_total = 0;
    for (int t = 0; t < 400; t++) {
        const int h = uni(30, 2200), w = uni(30, 3900);
        const int n = t % 37 == 0 ? 3000 : uni(0, 90);
        std::vector<float> lines((size_t)2 * (n ? n : 1));
        for (int i = 0; i < n; i++) {
            lines[2 * i] = (float)uni(-w, w + h);
            lines[2 * i + 1] = (float)(uni(0, 179) * 3.14159265358979 / 180);
            if (t % 37 == 0) lines[2 * i + 1] = (float)((i % 3) * 0.5 * 3.14159265358979 / 180);   // thousands of near-parallel lines
        }
        int32_t found = 0, update = 0, cen[8], ncen = 0, stats[2];
        int32_t hull[8] = { 10, 10, w - 10, 12, w - 8, h - 9, 9, h - 11 };
        const int rc = ck_boardfold_step(bf, h, w, uni(0, 2), lines.data(), n, t, t % 2 ? hull : nullptr, &found, &update, cen, &ncen, stats);
        if (rc != CK_OK && rc != CK_ERR_STATE) { std::printf("boardfold: rc %d\n", rc); return 1; }
        found_total += found;
        if (ncen < 0 || ncen > 4) { std::puts("boardfold: bad centre count"); return 1; }
    }
    {   // a real board: four bundles of near-duplicate lines, accumulated over four frames -> corners
        const float sides[4][2] = { { 100.f, 0.05f }, { 540.f, 0.08f }, { 60.f, 1.55f }, { 420.f, 1.62f } };
        ck_boardfold_reset(bf);
        for (int f = 0; f < 12; f++) {
            std::vector<float> lines;
            for (int k = 0; k < 4; k++)
                for (int d = 0; d < 1 + uni(0, 2); d++) { lines.push_back(sides[k][0] + (float)uni(-1, 1)); lines.push_back(sides[k][1]); }
            int32_t found = 0, update = 0, cen[8], ncen = 0, stats[2];
            if (ck_boardfold_step(bf, 480, 640, 0, lines.data(), (int)lines.size() / 2, f, nullptr, &found, &update, cen, &ncen, stats) != CK_OK) return 1;
            found_total += found;
        }
        if (!found_total) { std::puts("boardfold: the slanted board was not found"); return 1; }
    }
    ck_boardfold_destroy(bf);
    // ---- policy
    long requests = 0;
    for (int trial = 0; trial < 6; trial++) {
        ck_policy* p = nullptr;
        if (ck_policy_create(trial * 7, &p) != CK_OK) return 1;
        const int n = 700;
        std::vector<uint8_t> rl((size_t)n * 100), board(361, 0);
        std::vector<double> rc((size_t)n * 100);
        std::vector<int32_t> fg((size_t)n * 361), moves(3 * 722);
        for (auto& v : rl) v = (uint8_t)uni(0, 80);
        for (auto& v : rc) v = uni(0, 1000) / 1000.0;
        for (auto& v : fg) v = uni(0, 9) < 7 ? 0 : uni(0, 400);
        int32_t frame = 0, phase = 0, kind = 0, nm = 0;
        for (int guard = 0; guard < 100000; guard++) {
            if (ck_policy_run(p, n, 0, rl.data(), rc.data(), trial % 2 ? fg.data() : nullptr, board.data(), &frame, &phase, &kind,
                              moves.data(), 722, &nm) != CK_OK) { std::puts("policy: error"); return 1; }
            if (!kind) break;
            requests++;
            if (nm < 1 || nm > 722 || (kind == 1 && nm != 1)) { std::puts("policy: bad request"); return 1; }
            for (int i = 0; i < nm; i++) {
                const int col = moves[3 * i], r = moves[3 * i + 1], c = moves[3 * i + 2];
                if (col < 0 || col > 2 || r < 0 || r > 18 || c < 0 || c > 18) { std::puts("policy: bad move"); return 1; }
                board[(size_t)r * 19 + c] = (uint8_t)col;
            }
        }
        if (frame != n) { std::puts("policy: run did not finish"); return 1; }
        uint8_t tg[361], hc[361];
        int32_t he[361], flags[2];
        double cf[361];
        ck_policy_get_state(p, tg, hc, he, cf, flags);
        ck_policy_destroy(p);
    }
    std::printf("ordered halves: %ld hull vertices, %ld positive board frames, %ld policy requests, clean\n", hull_pts, found_total, requests);
    return 0;
}
