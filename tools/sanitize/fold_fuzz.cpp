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
    // ---- round 6: the folds over gathered records (ck_boardfold_run / ck_policy_run_records), records read through an order
    // table as the gather leaves them (rank after rank, a header row first), and the exact round(x, 10)
    long events = 0, rec_requests = 0;
    {
        const int W = 3, per = 40, n = W * per - 2;                     // the last two slots of the deal stay empty
        std::vector<ck_frame_record> rows((size_t)W * (per + 1));
        std::memset(rows.data(), 0, rows.size() * sizeof(ck_frame_record));
        std::vector<int32_t> order((size_t)n);
        const float sides[4][2] = { { 100.f, 0.05f }, { 540.f, 0.08f }, { 60.f, 1.55f }, { 420.f, 1.62f } };
        for (int f = 0; f < n; f++) {
            order[(size_t)f] = (f % W) * (per + 1) + 1 + f / W;
            ck_frame_record& r = rows[(size_t)order[(size_t)f]];
            r.status = uni(0, 9) < 8 ? CK_BOARD_LINES : uni(1, 2);
            r.n_lines = f % 17 == 0 ? 90 : uni(0, 12);                    // more lines found than a record holds: flagged, 64 kept
            r.flags = r.n_lines > CK_REC_LMAX ? CK_REC_LINES_CUT : 0;
            for (int i = 0; i < (r.n_lines < CK_REC_LMAX ? r.n_lines : CK_REC_LMAX); i++) {
                r.lines[i][0] = sides[i % 4][0] + (float)uni(-1, 1);
                r.lines[i][1] = sides[i % 4][1];
            }
            for (int k = 0; k < 100; k++) { r.region_label[k] = (uint8_t)uni(0, 80); r.region_conf[k] = uni(0, 1000) / 1000.0; }
        }
        ck_boardfold* bf2 = nullptr;
        if (ck_boardfold_create(&bf2) != CK_OK) return 1;
        for (int pass = 0; pass < 2; pass++) {
            int32_t k = 0, hold = 0, found = 0, update = 0, cen[8], ncen = 0, stats[2];
            long long counter = 0, seen_looked[2] = { 0, 0 };
            int32_t hull[8] = { 10, 10, 630, 12, 632, 470, 9, 468 };
            bool have_hull = false;
            for (int guard = 0; k < n && guard < 10000; guard++) {
                const int rc = ck_boardfold_run(bf2, 480, 640, rows.data(), pass ? order.data() : nullptr, pass ? n : (int)rows.size(), &k, &counter,
                                                &hold, seen_looked, have_hull ? hull : nullptr, pass ? 5 : -1, &found, &update, cen, &ncen, stats);
                if (rc == CK_ERR_STATE) { k++; counter++; continue; }     // the reference's IndexError: skip the frame, go on
                if (rc != CK_OK) { std::printf("boardfold_run: rc %d\n", rc); return 1; }
                if (update && ncen == 4) { std::memcpy(hull, cen, sizeof hull); have_hull = true; }
                if (found && (update || pass == 0)) hold = 5;
                events += found || update;
            }
            if (seen_looked[0] < seen_looked[1]) { std::puts("boardfold_run: looked at more frames than it saw"); return 1; }
        }
        ck_boardfold_destroy(bf2);
        ck_policy* p = nullptr;
        if (ck_policy_create(5, &p) != CK_OK) return 1;
        std::vector<int32_t> fg((size_t)n * 361), moves(3 * 722);
        std::vector<uint8_t> board(361, 0);
        for (auto& v : fg) v = uni(0, 9) < 7 ? 0 : uni(150, 400);
        int32_t frame = 0, phase = 0, kind = 0, nm = 0;
        for (int guard = 0; guard < 100000; guard++) {
            if (ck_policy_run_records(p, n, 0, rows.data(), order.data(), fg.data(), board.data(), &frame, &phase, &kind, moves.data(), 722,
                                      &nm) != CK_OK) { std::puts("policy_run_records: error"); return 1; }
            if (!kind) break;
            rec_requests++;
            for (int i = 0; i < nm; i++) board[(size_t)moves[3 * i + 1] * 19 + moves[3 * i + 2]] = (uint8_t)moves[3 * i];
        }
        if (frame != n) { std::puts("policy_run_records: run did not finish"); return 1; }
        ck_policy_destroy(p);
        const double probes[] = { 0.0, -0.0, 0.5, 0.49999999995, 1.00000000005, -1.0, 5e-324, 1e-300, 524287.99999999994, 524288.0, 1e22, -1e22 };
        for (double x : probes)
            if (ck_round10(x) != ck_round10_reference(x)) { std::printf("round10(%.17g): %.17g != %.17g\n", x, ck_round10(x), ck_round10_reference(x)); return 1; }
        for (int i = 0; i < 200000; i++) {
            const double x = (uni(-1000000, 1000000) + 0.5) / 1e10 * (i % 3 ? 1.0 : 1000.0);
            if (ck_round10(x) != ck_round10_reference(x)) { std::printf("round10(%.17g) differs\n", x); return 1; }
        }
    }
    std::printf("ordered halves: %ld hull vertices, %ld positive board frames, %ld policy requests, %ld fold events and %ld requests over gathered "
                "records, clean\n", hull_pts, found_total, requests, events, rec_requests);
    return 0;
}
