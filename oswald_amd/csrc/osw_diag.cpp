// oswald_amd/csrc/osw_diag.cpp -- kernel timing diagnostics of the -DOSW_DIAG build of the library (liboswald_hip_diag.so,
// `make -C oswald_amd/csrc diag`; loaded by tools/ on request, never by the product).  The shipped library does not contain
// this translation unit.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <utility>
#include <vector>

// t: per workgroup {start, end of phase 1, end (waves 0/1), end (waves 2/3)} in 100 MHz ticks, then (behind 4 * grid
// entries) the CU every workgroup ran on; reload_k: core-clock cycles / 1024 the waves of workgroup items spent in slice
// reloads and their barriers (OSWALD_HIP_DEBUG_TIMES)
void osw_diag_report_times(const unsigned long long *t, uint32_t grid, uint32_t reload_k)
{
    unsigned long long t0 = ~0ull, t1 = 0;
    for (uint32_t g = 0; g < grid; ++g) { if (t[g * 4]) t0 = std::min(t0, t[g * 4]); t1 = std::max(t1, std::max(t[g * 4 + 2], t[g * 4 + 3])); }
    const double span = (double)(t1 - t0) / 100.0; // us
    uint32_t hist_p1[10] = {0}, hist_end[10] = {0};
    double sum_end = 0;
    for (uint32_t g = 0; g < grid; ++g) {
        const double p1 = (double)(t[g * 4 + 1] - t0) / 100.0, e = (double)(std::max(t[g * 4 + 2], t[g * 4 + 3]) - t0) / 100.0;
        hist_p1[std::min(9, (int)(p1 / span * 10))]++;
        hist_end[std::min(9, (int)(e / span * 10))]++;
        sum_end += e;
    }
    fprintf(stderr, "[oswald_hip] DP launch span %.1f us over %u workgroups; mean finish at %.0f%% of span\n", span, grid, 100.0 * sum_end / grid / span);
    // waves x span in core-clock cycles (2.4 GHz nominal) against the cycles spent in the slice reloads of workgroup items
    const double wave_cycles = (double)grid * 4.0 * span * 2400.0;
    fprintf(stderr, "[oswald_hip]   workgroup items: slice reload + barrier waits %.3g cycles = %.1f%% of all wave time\n", (double)reload_k * 1024.0,
            100.0 * (double)reload_k * 1024.0 / wave_cycles);
    fprintf(stderr, "[oswald_hip]   phase-1 exits by decile:");
    for (int k = 0; k < 10; ++k) fprintf(stderr, " %u", hist_p1[k]);
    fprintf(stderr, "\n[oswald_hip]   finishes by decile:    ");
    for (int k = 0; k < 10; ++k) fprintf(stderr, " %u", hist_end[k]);
    fprintf(stderr, "\n");
    // per CU: when its LAST workgroup finished, and the time-integral of its resident workgroups
    std::vector<std::pair<unsigned long long, std::vector<double>>> cus; // (cu id, finish times of its workgroups)
    for (uint32_t g = 0; g < grid; ++g) {
        const unsigned long long cu = t[(size_t)grid * 4 + g];
        const double e = (double)(std::max(t[g * 4 + 2], t[g * 4 + 3]) - t0) / 100.0;
        auto it = std::find_if(cus.begin(), cus.end(), [&](const auto &x) { return x.first == cu; });
        if (it == cus.end()) { cus.push_back({cu, {}}); it = cus.end() - 1; }
        it->second.push_back(e);
    }
    uint32_t hist_last[10] = {0}, hist_n[8] = {0};
    double sum_last = 0, occ = 0;
    for (auto &c2 : cus) {
        const double last = *std::max_element(c2.second.begin(), c2.second.end());
        hist_last[std::min(9, (int)(last / span * 10))]++;
        hist_n[std::min<size_t>(7, c2.second.size())]++;
        sum_last += last;
        for (double e : c2.second) occ += e;
    }
    fprintf(stderr, "[oswald_hip]   %zu CUs; last finish per CU by decile:", cus.size());
    for (int k = 0; k < 10; ++k) fprintf(stderr, " %u", hist_last[k]);
    fprintf(stderr, "; mean last finish %.0f%% of span; workgroups per CU histogram:", 100.0 * sum_last / cus.size() / span);
    for (int k = 0; k < 8; ++k) fprintf(stderr, " %u", hist_n[k]);
    fprintf(stderr, "; mean resident workgroups per CU over the span %.2f\n", occ / cus.size() / span);
}
