// oswald_amd/host/oswald_main.cpp -- the `oswald` command line tool on MI355X.
//
// Keeps the reference's command line (reference host/src/arguments.c:14-35:
// -O preprocess|search|info, -i -o / -q -d -s -g -e -m -c -v -b -f -k -p -r),
// its preprocessed database format and its report (reference
// host/src/FPGAsearch.c:27-28, :60-65, :312-331).  The accelerator side of
// the search driver -- OpenCL bring-up, score-profile build, enqueue path --
// is replaced by calls into the C ABI of include/oswald_hip.h; this file is
// the mirror of fpga_search() (reference host/src/FPGAsearch.c:4-374) with
// GPUs where the reference has FPGAs.  `-m 0` (accelerator only) has no host compute path and fails loudly without a
// GPU; `-m 1` (hybrid, the reference's default: hybrid_search_*, host/src/HybridSearch.c) gives the host a share of
// the database measured on a test portion; `-m 2` (host only, BASELINE configs[0]) never touches a GPU.  The host
// kernel is this package's own (host_search.cpp).
#include <argp.h>
#include <algorithm>
#include <atomic>
#include <utility>
#include <sys/time.h>
#include <unistd.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <stdexcept>
#include <string>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

#include "oswald_hip.h"
#include "oswald_host.h"

namespace {

struct Options {
    const char *op = nullptr, *input = nullptr, *output = nullptr, *queries = nullptr, *db = nullptr;
    std::string submat = "blosum62", submat_name = "BLOSUM62";
    int open_gap = 10, extend_gap = 2, cpu_threads = 4, execution_mode = 1, cpu_vector_length = 16, cpu_block_size = 256;
    unsigned num_devices = 1;
    unsigned long top = 10, max_chunk_size = 134217728;
    double test_db_percentage = 0.01;
    int arg_count = 0;
};

const char *argp_doc =
    "\nOSWALD (MI355X build) accelerates Smith-Waterman protein database search; this build runs the search on AMD Instinct GPUs "
    "through HIP instead of the reference's Altera FPGAs";

int parse_opt(int key, char *arg, struct argp_state *state)
{
    Options *o = (Options *)state->input;
    switch (key) {
    case 'O':
        if (strcmp(arg, "preprocess") && strcmp(arg, "search") && strcmp(arg, "info"))
            argp_failure(state, 1, 0, "%s is not a valid option for execution.", arg);
        o->op = arg;
        break;
    case 'i': o->input = arg; break;
    case 'o': o->output = arg; break;
    case 'q': o->queries = arg; break;
    case 'd': o->db = arg; break;
    case 's': {
        if (!oswald::submat_by_name(arg)) argp_failure(state, 1, 0, "%s is not a valid option for substitution matrix.", arg);
        o->submat = arg;
        o->submat_name = arg;
        for (char &c : o->submat_name) c = (char)toupper((unsigned char)c);
        break;
    }
    case 'g':
        o->open_gap = atoi(arg);
        if (o->open_gap < 0 || o->open_gap > 255) argp_failure(state, 1, 0, "%d is not a valid option for gap open penalty.", o->open_gap);
        break;
    case 'e':
        o->extend_gap = atoi(arg);
        if (o->extend_gap < 0 || o->extend_gap > 127) argp_failure(state, 1, 0, "%d is not a valid option for gap extend penalty.", o->extend_gap);
        break;
    case 'm': {
        char *end = nullptr;
        const long v = !strcmp(arg, "host-only") ? 2 : strtol(arg, &end, 10);
        if ((end && (end == arg || *end)) || v < 0 || v > 2) argp_failure(state, 1, 0, "%s is not a valid option for execution mode.", arg);
        o->execution_mode = (int)v;
        break;
    }
    case 'c':
        o->cpu_threads = atoi(arg);
        if (o->cpu_threads < 0) argp_failure(state, 1, 0, "The number of host threads must be greater than 0.");
        break;
    case 'b':
        o->cpu_block_size = atoi(arg);
        if (o->cpu_block_size < 0) argp_failure(state, 1, 0, "The host block width must be greater than 0.");
        break;
    case 'v':
        o->cpu_vector_length = atoi(arg);
        if (o->cpu_vector_length != 16 && o->cpu_vector_length != 32) argp_failure(state, 1, 0, "%d is not a valid option for vector length of host.", o->cpu_vector_length);
        break;
    case 'f': {
        const int v = atoi(arg);
        if (v <= 0) argp_failure(state, 1, 0, "The number of GPUs must be greater than 0.");
        o->num_devices = (unsigned)v;
        break;
    }
    case 'k': {
        const long v = atol(arg);
        if (v <= 0) argp_failure(state, 1, 0, "The chunk size must be greater than 0.");
        o->max_chunk_size = (unsigned long)v;
        break;
    }
    case 'p':
        o->test_db_percentage = atof(arg);
        if (o->test_db_percentage <= 0 || o->test_db_percentage > 1) argp_failure(state, 1, 0, "The database percentage for testing must be between 0 and 1.");
        break;
    case 'r': {
        const long v = atol(arg);
        if (v < 0) argp_failure(state, 1, 0, "The number of scores to show must be greater than 0.");
        o->top = (unsigned long)v;
        break;
    }
    case ARGP_KEY_END:
        if (o->arg_count == 1) argp_failure(state, 1, 0, "Missing options");
        if (!o->op) argp_failure(state, 1, 0, "OSWALD execution option is required");
        else if (!strcmp(o->op, "preprocess")) {
            if (!o->input) argp_failure(state, 1, 0, "Input sequence filename is required");
            if (!o->output) argp_failure(state, 1, 0, "Output filename is required");
        } else if (!strcmp(o->op, "search")) {
            if (!o->db) argp_failure(state, 1, 0, "Database filename is required");
            if (!o->queries) argp_failure(state, 1, 0, "Query sequences filename is required");
        }
        break;
    default: break;
    }
    return 0;
}

double dwalltime()
{
    struct timeval tv;
    gettimeofday(&tv, nullptr);
    return tv.tv_sec + tv.tv_usec / 1000000.0;
}

// the reference's convention: any accelerator error prints and terminates (utils.c:256-262).  Off the main thread -- the
// threads that drive the devices, the hybrid mode's accelerator thread, OpenMP workers -- the process leaves through _exit():
// exit() would run the atexit handlers and static destructors (the HIP runtime's among them) under the feet of the other
// threads, which are inside runtime calls at that moment, and a clean message could end in a crash or a hang instead.
const std::thread::id g_main_thread = std::this_thread::get_id();
void check(int rc, const char *what)
{
    if (rc != 0) {
        fprintf(stderr, "OSWALD: %s failed: %s\n", what, oswald_hip_last_error());
        fflush(stdout);
        fflush(stderr);
        if (std::this_thread::get_id() != g_main_thread) _exit(EXIT_FAILURE);
        exit(EXIT_FAILURE);
    }
}

// The database's residues page-locked in place for the length of the search (oswald_hip_host_register): the uploads are then
// plain asynchronous DMA at the rate of the link (57 GB/s on the round-5 box) and their calls return at once; from pageable
// memory -- a private mapping of the group cache -- the runtime stages every copy through its own buffers on the caller's
// thread (18 - 26 GB/s, and an upload call holds the host for as long).  Registering the mapped 378 MB of a 1 M-sequence
// database takes 4 ms (tools/pin_probe.hip, profiles/r05_pin_probe.txt).  Before the clock, like the reference's aligned
// buffers (sequences.c:470-476).  OSWALD_NO_PIN=1 (A/B hook) leaves the memory pageable; so does a kernel that refuses
// (a message with OSWALD_DEBUG_PHASES, no error: uploads from pageable memory work).
struct PinnedResidues {
    std::vector<void *> held;
    explicit PinnedResidues(const oswald::Database &db)
    {
        if (getenv("OSWALD_NO_PIN")) return;
        for (const auto &r : oswald::residue_ranges(db)) {
            // (whole pages: the start of a chunk's own buffer need not be page-aligned)
            const uintptr_t a = (uintptr_t)r.first & ~(uintptr_t)4095, e = ((uintptr_t)r.first + r.second + 4095) & ~(uintptr_t)4095;
            if (oswald_hip_host_register((void *)a, (size_t)(e - a)) == 0) held.push_back((void *)a);
            else if (getenv("OSWALD_DEBUG_PHASES")) fprintf(stderr, "[oswald] residues stay pageable: %s\n", oswald_hip_last_error());
        }
    }
    ~PinnedResidues() { for (void *p : held) (void)oswald_hip_host_unregister(p); }
    PinnedResidues(const PinnedResidues &) = delete;
    PinnedResidues &operator=(const PinnedResidues &) = delete;
};

// devices 0 .. num_devices-1, or the list in OSWALD_DEVICE_IDS ("0,0": two context devices on one GPU; test hook)
int bring_up(const Options &o, oswald_hip_ctx **ctx)
{
    std::vector<int> ids;
    if (const char *e = getenv("OSWALD_DEVICE_IDS"))
        for (const char *p = e; *p;) { ids.push_back(atoi(p)); while (*p && *p != ',') ++p; if (*p) ++p; }
    return oswald_hip_init((int)o.num_devices, ids.size() == o.num_devices ? ids.data() : nullptr, ctx);
}

int do_preprocess(const Options &o)
{
    const double tick = dwalltime();
    const oswald::PreprocessStats st = oswald::preprocess_db(o.input, o.output, o.cpu_threads);
    printf("\nOSWALD v%s\n\n", oswald::kVersion);
    printf("Database file:\t\t\t %s\n", o.input);
    printf("Database size:\t\t\t%ld sequences (%ld residues) \n", (long)st.sequences, (long)st.residues);
    printf("Preprocessed database name:\t%s\n", o.output);
    printf("Preprocessing time:\t\t%lf seconds\n\n", dwalltime() - tick);
    return 0;
}

int do_info()
{
    int n = 0;
    check(oswald_hip_device_count(&n), "device discovery");
    if (n <= 0) { fprintf(stderr, "OSWALD: no GPU found.\n"); return -1; }
    oswald_hip_ctx *ctx = nullptr;
    check(oswald_hip_init(n, nullptr, &ctx), "device bring-up");
    char buf[2048];
    for (int d = 0; d < n; ++d) {
        check(oswald_hip_info(ctx, d, buf, sizeof buf), "device info");
        fputs(buf, stdout);
    }
    oswald_hip_finalize(ctx);
    return 0;
}

void print_header(const Options &o, const oswald::Database &db)
{
    printf("Database size:\t\t\t%ld sequences (%ld residues) \n", (long)db.sequences_count, (long)db.D);
    printf("Longest database sequence: \t%d residues\n", db.sequences_db_max_length);
    printf("Substitution matrix:\t\t%s\n", o.submat_name.c_str());
    printf("Gap open penalty:\t\t%d\n", o.open_gap);
    printf("Gap extend penalty:\t\t%d\n", o.extend_gap);
    printf("Query filename:\t\t\t%s\n", o.queries);
}

// query sections and footer of the report (reference FPGAsearch.c:312-331, HybridSearch.c:1218-1234)
void print_report(const Options &o, const oswald::Queries &q, const oswald::Database &db, const std::vector<std::vector<int32_t>> &top_s,
                  const std::vector<std::vector<uint64_t>> &top_i, time_t current_time, double work_time, double gcups_time)
{
    std::vector<uint64_t> wanted;
    for (const auto &ti : top_i) wanted.insert(wanted.end(), ti.begin(), ti.end());
    const std::vector<std::string> headers = oswald::load_database_headers_at(o.db, wanted);
    size_t hpos = 0;
    for (uint64_t i = 0; i < q.m.size(); ++i) {
        printf("\nQuery no.\t\t\t%d\n", (int)i + 1);
        printf("Query description: \t\t%s\n", q.titles[i].c_str() + 1);
        printf("Query length:\t\t\t%d residues\n", q.m[i]);
        printf("\nScore\tSequence description\n");
        for (size_t j = 0; j < top_s[i].size(); ++j) {
            const std::string &h = headers[hpos++];
            printf("%d\t%s\n", top_s[i][j], h.empty() ? "" : h.c_str() + 1);
        }
    }
    printf("\nSearch date:\t\t\t%s", ctime(&current_time));
    printf("Search time:\t\t\t%lf seconds\n", work_time);
    printf("Search speed:\t\t\t%.2lf GCUPS\n", (double)(q.Q * db.D) / (gcups_time * 1000000000));
    printf("CPU threads:\t\t\t%d\n", o.cpu_threads);
    printf("CPU vector length:\t\t%d\n", 16);
    printf("CPU block width:\t\t%d\n", o.cpu_block_size);
    printf("Number of FPGAs:\t\t%u\n", o.num_devices);
    printf("FPGA vector length:\t\t%d\n", oswald::kFpgaVectorLength);
    printf("FPGA block width:\t\t%d\n", oswald::kFpgaBlockWidth);
    printf("Max. chunk size in FPGA:\t%ld bytes\n", (long)o.max_chunk_size);
    // (not part of the reference's report, FPGAsearch.c:327-331: only on request)
    if (getenv("OSWALD_REPORT_ACCELERATOR"))
        printf("Accelerator:\t\t\t%u x AMD Instinct GPU via HIP (the \"FPGA\" lines above describe the input layout)\n", o.num_devices);
}

void tops_from_table(const Options &o, const oswald::Database &db, const std::vector<int32_t> &scores, uint64_t nq,
                     std::vector<std::vector<int32_t>> &top_s, std::vector<std::vector<uint64_t>> &top_i)
{
    const uint64_t W = oswald::kFpgaVectorLength;
    top_s.assign(nq, {});
    top_i.assign(nq, {});
    for (uint64_t i = 0; i < nq; ++i) oswald::top_scores(scores.data() + i * db.vect_sequences_count * W, db.sequences_count, o.top, top_s[i], top_i[i]);
}

// -k adapted to what the devices can hold BEFORE the database is cut, in every mode that uses them -- the reference does it in
// init() (utils.c:162-168, called from main.c:46 whatever -m says)
void clamp_chunk_size(Options &o, oswald_hip_ctx *ctx, uint64_t nq)
{
    for (unsigned d = 0; d < o.num_devices; ++d) {
        uint64_t fits = 0;
        check(oswald_hip_max_chunk_size(ctx, (int)d, (uint32_t)nq, oswald::kMaxSequenceLength, &fits), "chunk size limit");
        if (fits < o.max_chunk_size) o.max_chunk_size = fits;
    }
}

// -m 2: every group on the host cores; no accelerator call at all.
// -c as asked for, but no more threads than the process may keep busy (see usable_cpus: beyond a container's CPU quota a team
// is throttled for most of every scheduling period)
int team_size(const Options &o, unsigned reserve = 0)
{
    const unsigned hw = oswald::usable_cpus();
    const int room = hw > reserve ? (int)(hw - reserve) : 1;
    const int want = std::max(o.cpu_threads, 1);
    if (want > room) fprintf(stderr, "oswald: %d host threads (-c %d asked for; the process may keep %u hardware threads busy)\n", room, want, hw);
    return std::min(want, room);
}

int do_search_host_only(Options &o)
{
    const time_t current_time = time(nullptr);
    printf("\nOSWALD v%s \n\n", oswald::kVersion);
    printf("Database file:\t\t\t%s\n", o.db);
    oswald::Queries q = oswald::load_query_sequences(o.queries);
    oswald::Database db = oswald::assemble_multiple_chunks_db(o.db, oswald::kFpgaVectorLength, o.max_chunk_size, 1);
    const uint64_t nq = q.m.size(), W = oswald::kFpgaVectorLength;
    print_header(o, db);
    if (db.sequences_count < o.top) o.top = db.sequences_count;
    std::vector<int32_t> scores(nq * db.vect_sequences_count * W, 0);
    const int threads = team_size(o);
    const double tick = dwalltime();
    for (const oswald::Chunk &c : db.chunks)
        oswald::host_search_groups(q, c, 0, c.n.size(), (int)W, oswald::submat_by_name(o.submat), o.open_gap, o.extend_gap, threads, scores.data(),
                                   db.vect_sequences_count * W, c.accum * W, o.cpu_vector_length, nullptr, nullptr, o.cpu_block_size);
    const double work_time = dwalltime() - tick;
    std::vector<std::vector<int32_t>> top_s;
    std::vector<std::vector<uint64_t>> top_i;
    tops_from_table(o, db, scores, nq, top_s, top_i);
    print_report(o, q, db, top_s, top_i, current_time, work_time, work_time);
    return 0;
}

// A run of whole groups [g0, g1) of a chunk as a chunk of its own (displacements rebased): what the hybrid mode
// hands to the accelerator.
struct GroupRun {
    const uint8_t *b;
    uint64_t bytes;
    const uint16_t *n;
    std::vector<uint32_t> disp;
    uint64_t first_group; // in the whole database
};
GroupRun group_run(const oswald::Chunk &c, uint64_t g0, uint64_t g1)
{
    GroupRun r;
    r.b = c.b + c.disp[g0];
    r.n = c.n.data() + g0;
    r.disp.resize(g1 - g0);
    for (uint64_t g = g0; g < g1; ++g) r.disp[g - g0] = c.disp[g] - c.disp[g0];
    const uint64_t end = g1 < c.n.size() ? c.disp[g1] : c.b_size;
    r.bytes = end - c.disp[g0];
    r.first_group = c.accum + g0;
    return r;
}

// -m 1: the reference's hybrid scheme (hybrid_search_*, HybridSearch.c:124-228, :620-631; assemble_db_chunks,
// sequences.c:828-1094): both sides search a test portion of the database (the first -p of its groups) to measure
// their speeds, the rest is divided in that proportion -- the accelerator takes the groups that follow the test
// portion, the host the longest sequences at the end -- and both work at the same time.
// This is the reference's STATIC division with the whole score table brought to the host, as the reference does it; it
// serves -r > 1024 (top lists longer than the devices select).  do_search_hybrid below is what -m 1 normally runs.
int do_search_hybrid_static(Options &o)
{
    const time_t current_time = time(nullptr);
    printf("\nOSWALD v%s \n\n", oswald::kVersion);
    printf("Database file:\t\t\t%s\n", o.db);
    oswald::Queries q = oswald::load_query_sequences(o.queries);
    oswald_hip_ctx *ctx = nullptr;
    check(bring_up(o, &ctx), "device bring-up");
    clamp_chunk_size(o, ctx, q.m.size());
    oswald::Database db = oswald::assemble_multiple_chunks_db(o.db, oswald::kFpgaVectorLength, o.max_chunk_size, o.num_devices);
    PinnedResidues pinned(db);
    const uint64_t nq = q.m.size(), W = oswald::kFpgaVectorLength, G = db.vect_sequences_count, row = G * W;
    print_header(o, db);
    if (db.sequences_count < o.top) o.top = db.sequences_count;
    const int8_t *sm = oswald::submat_by_name(o.submat);
    std::vector<int32_t> scores(nq * row, 0);
    check(oswald_hip_set_scoring(ctx, sm, o.open_gap, o.extend_gap, 0), "scoring setup");
    check(oswald_hip_set_queries(ctx, q.a.data(), q.Q, q.m.data(), q.a_disp.data(), (uint32_t)nq), "query upload");
    auto padded = [&](uint64_t g0, uint64_t g1) { // padded residues of database groups [g0, g1)
        uint64_t v = 0;
        for (const oswald::Chunk &c : db.chunks)
            for (uint64_t g = std::max(g0, c.accum); g < std::min<uint64_t>(g1, c.accum + c.n.size()); ++g) v += (uint64_t)c.n[g - c.accum] * W;
        return v;
    };
    const int static_team = team_size(o, 2 * o.num_devices + 2);
    // searches database groups [g0, g1) on the accelerator(s), chunk by chunk, into the score table
    auto gpu_groups = [&](uint64_t g0, uint64_t g1) {
        std::vector<GroupRun> runs;
        for (const oswald::Chunk &c : db.chunks) {
            const uint64_t a0 = std::max(g0, c.accum), a1 = std::min<uint64_t>(g1, c.accum + c.n.size());
            if (a0 < a1) runs.push_back(group_run(c, a0 - c.accum, a1 - c.accum));
        }
        std::vector<std::vector<int32_t>> tmp(o.num_devices);
        for (size_t k = 0; k < runs.size(); k += o.num_devices) { // run k + d of a round goes to device d (FPGAsearch.c:132-138)
            const size_t active = std::min<size_t>(o.num_devices, runs.size() - k);
            for (size_t d = 0; d < active; ++d) {
                const GroupRun &r = runs[k + d];
                tmp[d].assign(nq * r.disp.size() * W, 0);
                check(oswald_hip_search_chunk_async(ctx, (int)d, r.b, r.bytes, r.n, r.disp.data(), (uint32_t)r.disp.size(), (uint32_t)W, tmp[d].data()),
                      "chunk search");
            }
            check(oswald_hip_wait(ctx, -1), "wait");
            for (size_t d = 0; d < active; ++d) {
                const GroupRun &r = runs[k + d];
                const size_t cols = r.disp.size() * W;
                for (uint64_t qi = 0; qi < nq; ++qi) memcpy(scores.data() + qi * row + r.first_group * W, tmp[d].data() + qi * cols, cols * sizeof(int32_t));
            }
        }
    };
    // database groups [g0, g1) on the host cores, into `dst` (rows of `dst_row` scores, group `dst_g0` in column 0)
    auto cpu_groups = [&](uint64_t g0, uint64_t g1, int32_t *dst, uint64_t dst_row, uint64_t dst_g0, const std::atomic<bool> *cancel = nullptr,
                          std::atomic<uint64_t> *done = nullptr) {
        for (const oswald::Chunk &c : db.chunks) {
            const uint64_t a0 = std::max(g0, c.accum), a1 = std::min<uint64_t>(g1, c.accum + c.n.size());
            if (a0 < a1) oswald::host_search_groups(q, c, a0 - c.accum, a1 - c.accum, (int)W, sm, o.open_gap, o.extend_gap, static_team, dst, dst_row, (a0 - dst_g0) * W, o.cpu_vector_length,
                                                    cancel, done, o.cpu_block_size);
        }
    };
    // Test portion.  The host searches the first -p of the groups, as in the reference.  The accelerator's speed
    // cannot be read off so small a portion (a launch has a fixed cost of about a millisecond: 1 % of a 100 000-sequence
    // database would rate an MI355X at 60 GCUPS): it goes on from the start of the database in portions of doubling
    // size until one takes 20 ms, and that one is its measurement.  Everything it computed on the way is final.
    // (the reference sizes the test portion in residues, test_chunk_size = D x p, sequences.c:680: the shortest groups that hold them)
    uint64_t test_groups = 0;
    {
        const double want = o.test_db_percentage * (double)db.vD;
        double have = 0;
        for (const oswald::Chunk &c : db.chunks)
            for (uint64_t g = 0; g < c.n.size() && have < want; ++g) { have += (double)c.n[g] * W; ++test_groups; }
        test_groups = std::min<uint64_t>(G, std::max<uint64_t>(1, test_groups));
    }
    // Both sides run their test at the same time, as in the reference (two `omp single nowait` blocks,
    // HybridSearch.c:124-228), which is why the report charges max(test_fpga_time, test_cpu_time), :1227.  The host's
    // test scores go to a table of their own and are dropped: the accelerator fills the same columns for good.  The
    // host's test is called off when the accelerator's is over (1 % of a 1 M-sequence database keeps four host threads
    // busy for a quarter of a minute; an MI355X is rated after 50 ms and needs 0.4 s for the whole database): -p is an
    // upper bound here, and the host is rated on the (group, query) pairs it did finish (at least one).
    double test_gpu_time = 0, test_cpu_time = 0, gpu_gcups = 0;
    uint64_t gpu_done = 0;
    const double tick_test = dwalltime();
    std::atomic<bool> call_off{false}, host_done{false};
    std::atomic<uint64_t> host_test_cells{0};
    {
        std::vector<int32_t> test_scores(nq * test_groups * W, 0);
        std::thread host_test([&] {
            const double t = dwalltime();
            cpu_groups(0, test_groups, test_scores.data(), test_groups * W, 0, &call_off, &host_test_cells);
            test_cpu_time = dwalltime() - t;
            host_done.store(true);
        });
        for (uint64_t n = test_groups; gpu_done < G; n *= 2) {
            const uint64_t g1 = std::min<uint64_t>(G, gpu_done + n);
            const double t = dwalltime();
            gpu_groups(gpu_done, g1);
            const double dt = dwalltime() - t;
            gpu_gcups = q.Q * (double)padded(gpu_done, g1) / (dt * 1e9);
            test_gpu_time += dt;
            gpu_done = g1;
            if (dt >= 0.02) break;
        }
        // (at least one finished (group, query) to rate the host on)
        while (host_test_cells.load() == 0 && !host_done.load()) {
            struct timespec ts = {0, 1000000};
            nanosleep(&ts, nullptr);
            if (dwalltime() - (tick_test + test_gpu_time) > 60.0) break; // (a host that cannot finish one group in a minute is rated 0)
        }
        call_off.store(true);
        host_test.join();
    }
    const double cpu_gcups = (double)host_test_cells.load() / (std::max(test_cpu_time, 1e-9) * 1e9); // every (group, query) the host finished
    printf("Test DB percentage:\t\t%.4lf%% \n", o.test_db_percentage);
    printf("CPU estimated speed:\t\t%.2lf GCUPS\n", cpu_gcups);
    printf("FPGA estimated speed:\t\t%.2lf GCUPS\n", gpu_gcups);
    const double gpu_pow = gpu_gcups / (gpu_gcups + cpu_gcups);
    // the accelerator takes groups [gpu_done, split) = gpu_pow of the remaining padded residues, the host the rest
    uint64_t split = gpu_done;
    {
        const double want = gpu_pow * (double)padded(gpu_done, G);
        double have = 0;
        for (const oswald::Chunk &c : db.chunks)
            for (uint64_t g = std::max(gpu_done, c.accum); g < c.accum + c.n.size() && have < want; ++g) { have += (double)c.n[g - c.accum] * W; split = g + 1; }
    }
    const double tick = dwalltime();
    {
        std::thread gpu([&] { gpu_groups(gpu_done, split); });
        cpu_groups(split, G, scores.data(), row, 0);
        gpu.join();
    }
    const double work_time = dwalltime() - tick;
    oswald_hip_finalize(ctx);
    std::vector<std::vector<int32_t>> top_s;
    std::vector<std::vector<uint64_t>> top_i;
    tops_from_table(o, db, scores, nq, top_s, top_i);
    print_report(o, q, db, top_s, top_i, current_time, work_time, work_time + std::max(test_gpu_time, test_cpu_time));
    return 0;
}

// -m 1 as it normally runs (second session of round 4).  The same test portion, the same three report lines, the same
// roles -- the accelerator works up from the shortest sequences, the host down from the longest, at the same time -- but
// the boundary between them is not fixed by the test: both sides TAKE work until they meet.  The accelerator takes
// pieces of up to -k bytes from the front (never more than its rated share of what is left, so that its last pieces get
// small), the host batches of about ten milliseconds from the back (and none once the accelerator would be through with
// everything else sooner).  A static split stands and falls with the two ratings: on an MI355X the accelerator is rated
// on a 20 ms portion at 0.75 - 0.9 of its real speed and the host on a handful of groups, and the side that was given too
// much keeps the other waiting -- 1 M sequences took 0.38 s with 16 host threads and 0.74 s with 64 where the
// accelerator alone needs 0.35 s.  The accelerator side is the pipeline of the accelerator-only mode: uploads a piece
// ahead, top lists selected on the devices (oswald_hip_topr), no score table brought back; the host's scores go to the
// table and its top candidates are merged with the devices' by the reference's rule (utils.c:3-86).
int do_search_hybrid(Options &o)
{
    if (o.top > 1024) return do_search_hybrid_static(o);
    // OSWALD_HYBRID_TEST_SPLIT=1 (test hook): a database of any size is shared out -- the accelerator's first piece is half of the
    // rest, no piece swallows a sliver, and the host takes batches whatever the ratings say (tests/test_gpu_cli.py)
    const bool force_split = getenv("OSWALD_HYBRID_TEST_SPLIT") != nullptr;
    const time_t current_time = time(nullptr);
    printf("\nOSWALD v%s \n\n", oswald::kVersion);
    printf("Database file:\t\t\t%s\n", o.db);
    oswald::Queries q = oswald::load_query_sequences(o.queries);
    oswald_hip_ctx *ctx = nullptr;
    check(bring_up(o, &ctx), "device bring-up");
    clamp_chunk_size(o, ctx, q.m.size());
    oswald::Database db = oswald::assemble_multiple_chunks_db(o.db, oswald::kFpgaVectorLength, o.max_chunk_size, 1);
    PinnedResidues pinned(db);
    const uint64_t nq = q.m.size(), W = oswald::kFpgaVectorLength, G = db.vect_sequences_count, row = G * W;
    print_header(o, db);
    if (db.sequences_count < o.top) o.top = db.sequences_count;
    const int8_t *sm = oswald::submat_by_name(o.submat);
    // The host's team leaves a hardware thread to every thread that drives an accelerator (and one to the rest of the process),
    // out of the hardware threads the process may really keep busy (oswald::usable_cpus: its affinity mask AND its cgroup's CPU
    // bandwidth).  Round 4 saw the accelerator's 10-ms test take 130 - 190 ms with -c 96 ... 128 "although a quarter of the cores
    // were idle": the box gives its container 16 CPUs' worth of a 256-thread host, and a team that burns the quota of a 100-ms
    // period in its first 12 ms is throttled -- every thread of the process, the one that drives the accelerator included --
    // for the rest of it (round 5, profiles/r05_cli_hybrid.txt).  The report prints the -c that was asked for.
    int host_threads = std::max(o.cpu_threads, 1);
    {
        const unsigned hw = oswald::usable_cpus();
        const int room = hw > 2 * o.num_devices + 2 ? (int)(hw - 2 * o.num_devices - 2) : 1; // (a driving thread and a runtime thread per accelerator, two for the rest)
        if (host_threads > room) {
            fprintf(stderr, "oswald: hybrid mode runs the host part on %d threads (-c %d asked for; the process may keep %u hardware threads busy, %u of them drive the accelerators)\n", room, host_threads, hw, o.num_devices);
            host_threads = room;
        }
    }
    const uint64_t host_min_groups = std::max<uint64_t>(1, (2 * (uint64_t)host_threads + nq - 1) / std::max<uint64_t>(nq, 1)); // >= two (group, query) cells per host thread
    std::vector<int32_t> scores(nq * row, 0); // the host's columns only
    {   // the chunk slots BEFORE the clock, where the reference's hybrid driver creates its device buffers (HybridSearch.c:79-90, its clocks start at :124 /
        // :633; fpga_search creates them behind its tick, FPGAsearch.c:80-96, and so does `-m 0` here): a piece is a run of groups of one chunk
        uint32_t mg = 0;
        for (const oswald::Chunk &c : db.chunks) mg = std::max<uint32_t>(mg, (uint32_t)c.n.size());
        if (db.max_chunk_vD) check(oswald_hip_reserve_chunks(ctx, -1, db.max_chunk_vD, mg, (uint32_t)W, (uint32_t)nq, 3), "device buffers");
    }
    check(oswald_hip_set_scoring(ctx, sm, o.open_gap, o.extend_gap, 0), "scoring setup");
    check(oswald_hip_set_queries(ctx, q.a.data(), q.Q, q.m.data(), q.a_disp.data(), (uint32_t)nq), "query upload");
    check(oswald_hip_topr_begin(ctx, (uint32_t)o.top), "top scores");
    // padded residues before every group of the database, and the chunk a group lies in
    std::vector<uint64_t> pre(G + 1, 0);
    for (const oswald::Chunk &c : db.chunks)
        for (uint64_t g = 0; g < c.n.size(); ++g) pre[c.accum + g + 1] = (uint64_t)c.n[g] * W;
    for (uint64_t g = 0; g < G; ++g) pre[g + 1] += pre[g];
    auto chunk_of = [&](uint64_t g) -> const oswald::Chunk & {
        for (const oswald::Chunk &c : db.chunks) if (g >= c.accum && g < c.accum + c.n.size()) return c;
        return db.chunks.back();
    };
    auto cells = [&](uint64_t g0, uint64_t g1) { return (double)(pre[g1] - pre[g0]) * (double)q.Q; };
    // the accelerator side of a range of groups inside ONE chunk: upload (asynchronous), index, search; the run is kept until the device is through
    struct Live { GroupRun run; int handle; };
    auto gpu_upload = [&](int dev, uint64_t g0, uint64_t g1, std::vector<std::unique_ptr<Live>> &keep) {
        const oswald::Chunk &c = chunk_of(g0);
        std::unique_ptr<Live> l(new Live{group_run(c, g0 - c.accum, g1 - c.accum), -1});
        check(oswald_hip_chunk_upload_async(ctx, dev, l->run.b, l->run.bytes, l->run.n, l->run.disp.data(), (uint32_t)l->run.disp.size(), (uint32_t)W, &l->handle), "chunk upload");
        const uint64_t first = g0 * W, last = std::min<uint64_t>(db.sequences_count, g1 * W);
        check(oswald_hip_chunk_set_index(ctx, dev, l->handle, (uint32_t)first, (uint32_t)(last > first ? last - first : 0), nullptr), "chunk index");
        keep.push_back(std::move(l));
        return keep.back()->handle;
    };
    // (the slot is given back separately, AFTER the next piece's upload has been queued: the release returns when the piece's own
    // upload has landed, and the next piece must be on the link by then)
    auto gpu_search = [&](int dev, int handle) { check(oswald_hip_chunk_search(ctx, dev, handle, nullptr), "chunk search"); };
    auto gpu_release = [&](int dev, int handle) { check(oswald_hip_chunk_release(ctx, dev, handle), "chunk release"); };
    auto cpu_groups = [&](uint64_t g0, uint64_t g1, int32_t *dst, uint64_t dst_row, uint64_t dst_g0, const std::atomic<bool> *cancel = nullptr,
                          std::atomic<uint64_t> *done = nullptr) {
        for (const oswald::Chunk &c : db.chunks) {
            const uint64_t a0 = std::max(g0, c.accum), a1 = std::min<uint64_t>(g1, c.accum + c.n.size());
            if (a0 < a1) oswald::host_search_groups(q, c, a0 - c.accum, a1 - c.accum, (int)W, sm, o.open_gap, o.extend_gap, host_threads, dst, dst_row, (a0 - dst_g0) * W, o.cpu_vector_length,
                                                    cancel, done, o.cpu_block_size);
        }
    };
    // Test portion: the host on the first -p of the groups, called off when the accelerator's rating is over (as in
    // do_search_hybrid_static); the accelerator on one portion from the start of the database -- what it computes there is
    // final: its top candidates are in the devices' running lists
    uint64_t test_groups = 0;
    {
        const double want = o.test_db_percentage * (double)db.vD;
        while (test_groups < G && (double)pre[test_groups] < want) ++test_groups;
        test_groups = std::min<uint64_t>(G, std::max<uint64_t>(1, test_groups));
    }
    double test_gpu_time = 0, test_cpu_time = 0, gpu_gcups = 0;
    uint64_t gpu_done = 0, spec_g1 = 0;
    int spec_handle = -1;
    {   // (the host's thread team exists before the clock starts: creating 128 threads takes a tenth of a second)
        int team = 0;
#pragma omp parallel num_threads(host_threads) reduction(+ : team)
        team += 1;
        (void)team;
    }
    const double tick_test = dwalltime();
    std::atomic<bool> call_off{false}, host_done{false};
    std::atomic<uint64_t> host_test_cells{0};
    std::vector<std::unique_ptr<Live>> keep0;
    {
        // The host's test runs on THIS thread (its thread team exists: see above; a team is per master thread), the
        // accelerator's on a thread of its own, which calls the host's test off when its rating is over and the host has
        // finished at least one (group, query).
        std::vector<int32_t> test_scores(nq * test_groups * W, 0);
        std::thread gpu_test([&] {
            // ONE portion: the test portion or the shortest groups that hold 8 MiB of padded residues (a twentieth of a small database), whichever is larger (the
            // rating only sizes the accelerator's last pieces and the host's batches here -- the boundary between the two sides
            // is found by the work itself --, so it need not be exact; a launch has a fixed cost of about a millisecond, and
            // 8 MiB against 20 queries is ~8 ms of work)
            uint64_t g1 = test_groups;
            const uint64_t cal_bytes = std::min<uint64_t>(8ull << 20, pre[G] / 20); // (a twentieth of a small database at most)
            while (g1 < G && pre[g1] < cal_bytes) ++g1;
            const double t = dwalltime();
            std::vector<int> test_handles;
            const bool tphases = getenv("OSWALD_DEBUG_PHASES") != nullptr;
            auto stamp = [&](const char *what) { if (tphases) fprintf(stderr, "[oswald] accelerator test thread at %7.2f ms: %s\n", (dwalltime() - tick_test) * 1e3, what); };
            stamp("started");
            for (uint64_t g = gpu_done; g < g1;) { // (a portion may span chunks)
                const oswald::Chunk &c = chunk_of(g);
                const uint64_t e = std::min<uint64_t>(g1, c.accum + c.n.size());
                test_handles.push_back(gpu_upload(0, g, e, keep0));
                stamp("test portion's upload queued");
                gpu_search(0, test_handles.back());
                stamp("test portion's search queued");
                g = e;
            }
            // the accelerator's first piece of the rest comes in beside the test search: what follows the test portion in its
            // chunk, nine tenths of what is left at most (whatever the ratings turn out to be, that much is the accelerator's) -- all
            // of it when that is less than 64 MiB
            if (g1 < G) {
                const oswald::Chunk &c = chunk_of(g1);
                uint64_t e = g1 + 1;
                const double rest = (double)(pre[G] - pre[g1]);
                const double cap = force_split ? 0.5 * rest : rest <= 64.0 * 1048576.0 ? rest : std::min((double)o.max_chunk_size, 0.9 * rest); // (a small rest is one piece: a second launch would cost more than the host can give)
                while (e < c.accum + c.n.size() && (double)(pre[e + 1] - pre[g1]) <= cap) ++e;
                spec_g1 = e;
                spec_handle = gpu_upload(0, g1, e, keep0);
                // ... and its search is queued behind the test portion's at once: the device goes from one into the other, while
                // this thread waits for the test portion alone (round 4 waited for the device, then planned the piece -- 2.5 ms of
                // host work at 1 M sequences -- with the device idle)
                gpu_search(0, spec_handle);
                stamp("first piece of the rest queued, upload and search");
            }
            for (int h : test_handles) check(oswald_hip_chunk_wait(ctx, 0, h), "wait for the test portion");
            stamp("the device is through with the test portion");
            for (int h : test_handles) gpu_release(0, h);
            const double dt = dwalltime() - t;
            gpu_gcups = cells(gpu_done, g1) / (std::max(dt, 1e-9) * 1e9);
            test_gpu_time += dt;
            gpu_done = g1;
            while (host_test_cells.load() == 0 && !host_done.load()) { // (at least one finished (group, query) to rate the host on)
                struct timespec ts = {0, 200000};
                nanosleep(&ts, nullptr);
                if (dwalltime() - (tick_test + test_gpu_time) > 60.0) break; // (a host that cannot finish one group in a minute is rated 0)
            }
            call_off.store(true);
        });
        const double t = dwalltime();
        // (in batches, so that the team sleeps for a moment every millisecond or so: one long parallel region
        // on every hardware thread keeps the runtime's own threads -- which complete the accelerator's calls -- off the cores)
        const uint64_t test_batch = 16 * host_min_groups; // (~30 (group, query) cells per thread: the team's wake-up is small against it)
        for (uint64_t g = 0; g < test_groups && !call_off.load(); g += test_batch)
            cpu_groups(g, std::min<uint64_t>(test_groups, g + test_batch), test_scores.data(), test_groups * W, 0, &call_off, &host_test_cells);
        test_cpu_time = dwalltime() - t;
        host_done.store(true);
        gpu_test.join();
    }
    const double cpu_gcups = (double)host_test_cells.load() / (std::max(test_cpu_time, 1e-9) * 1e9);
    printf("Test DB percentage:\t\t%.4lf%% \n", o.test_db_percentage);
    printf("CPU estimated speed:\t\t%.2lf GCUPS\n", cpu_gcups);
    printf("FPGA estimated speed:\t\t%.2lf GCUPS\n", gpu_gcups);

    // The rest: groups [front, back) are nobody's yet.
    std::mutex mx;
    double cpu_live = cpu_gcups; // the host's speed on what it is working on now (the test rated it on the SHORTEST sequences)
    uint64_t front = spec_handle >= 0 ? spec_g1 : gpu_done, back = G;
    auto take_gpu = [&](uint64_t &g0, uint64_t &g1) {
        std::lock_guard<std::mutex> lk(mx);
        if (front >= back) return false;
        const double left = (double)(pre[back] - pre[front]);
        // (what is left of a last few MiB is one piece: the accelerator is through with it in milliseconds, and every launch costs one)
        const double pow_now = gpu_gcups / std::max(gpu_gcups + cpu_live, 1e-9); // (the host's rating follows what it is doing now)
        double want = std::min((double)o.max_chunk_size, std::max(1.0, pow_now * left / std::max(1u, o.num_devices)));
        // no sliver behind the last piece (it would cost a launch of its own): what the accelerator does in ~10 ms goes with it
        const double sliver = std::max(4.0 * 1048576.0, gpu_gcups * 1e9 * 0.010 / std::max((double)q.Q, 1.0));
        if (force_split) want = std::max(1.0, 0.5 * want);
        else if (left - want <= sliver && left <= (double)o.max_chunk_size) want = left;
        const oswald::Chunk &c = chunk_of(front);
        const uint64_t end = std::min<uint64_t>(back, c.accum + c.n.size());
        g0 = front;
        g1 = front + 1;
        while (g1 < end && (double)(pre[g1 + 1] - pre[g0]) <= want) ++g1;
        front = g1;
        return true;
    };
    auto take_cpu = [&](uint64_t &g0, uint64_t &g1) {
        std::lock_guard<std::mutex> lk(mx);
        if (front >= back || cpu_live <= 0) return false;
        const double target = force_split ? 0.0 : cpu_live * 1e9 * 0.010; // cells of ~10 ms of host work
        g1 = back;
        g0 = back;
        while (g0 > front && (g1 - g0 < host_min_groups || cells(g0, g1) < target)) --g0;
        // ... unless the accelerator would be through with everything else before the host is through with this batch -- with a margin
        // of 2.5 on the host's time: since its kernel starts in int8 (round 5) a batch of the longest sequences may take twice what the
        // rating says (groups that reach 127 are redone in int16), and a batch that overruns the end by 5 ms costs a 100 000-sequence
        // search 13 % (seen twice in 25 runs; the host's whole share is 1 %)
        if (!force_split && gpu_gcups > 0 && cells(front, g0) / gpu_gcups < 2.5 * cells(g0, g1) / cpu_live) return false;
        back = g0;
        return true;
    };
    const double tick = dwalltime();
    const bool phases = getenv("OSWALD_DEBUG_PHASES") != nullptr;
    uint64_t host_first = G; // the host's groups: [host_first, G)
    std::vector<std::vector<std::unique_ptr<Live>>> keep(o.num_devices);
    {
        std::vector<std::thread> devs;
        for (unsigned d = 0; d < o.num_devices; ++d)
            devs.emplace_back([&, d] {
                uint64_t g0, g1;
                int cur = d == 0 ? spec_handle : -1; // (device 0 holds the piece that came in beside the test search ...
                bool queued = cur >= 0;               //  ... and whose search is queued already)
                if (cur < 0) {
                    if (!take_gpu(g0, g1)) return;
                    cur = gpu_upload((int)d, g0, g1, keep[d]);
                }
                for (;;) { // search the piece in hand, then bring the next one in beside it
                    const double t0 = dwalltime();
                    if (!queued) gpu_search((int)d, cur);
                    queued = false;
                    const double t1 = dwalltime();
                    const int searched = cur;
                    const bool more = take_gpu(g0, g1);
                    if (more) cur = gpu_upload((int)d, g0, g1, keep[d]);
                    gpu_release((int)d, searched);
                    if (!more) break;
                    if (phases) fprintf(stderr, "[oswald] device %u at %.1f ms: search + release calls %.1f ms, next piece %.1f MiB queued in %.1f ms\n", d, (t0 - tick) * 1e3, (t1 - t0) * 1e3,
                                        (double)(pre[g1] - pre[g0]) / 1048576.0, (dwalltime() - t1) * 1e3);
                }
                if (phases) fprintf(stderr, "[oswald] device %u has queued its last piece at %.1f ms\n", d, (dwalltime() - tick) * 1e3);
            });
        uint64_t g0, g1;
        while (take_cpu(g0, g1)) {
            const double t = dwalltime();
            cpu_groups(g0, g1, scores.data(), row, 0);
            host_first = g0;
            const double rate = cells(g0, g1) / (std::max(dwalltime() - t, 1e-9) * 1e9);
            std::lock_guard<std::mutex> lk(mx);
            cpu_live = 0.5 * (cpu_live + rate);
        }
        if (phases) fprintf(stderr, "[oswald] host done at %.1f ms: groups %llu .. %llu of %llu, %.2f %% of the padded residues, live rate %.1f GCUPS\n", (dwalltime() - tick) * 1e3,
                            (unsigned long long)host_first, (unsigned long long)G, (unsigned long long)G, 100.0 * (double)(pre[G] - pre[host_first]) / (double)pre[G], cpu_live);
        for (std::thread &t : devs) t.join();
    }
    // the devices' lists, the host's candidates, merged by the reference's rule
    std::vector<int32_t> cand_s(nq * 2 * o.top, -1);
    std::vector<uint32_t> cand_i(nq * 2 * o.top, 0);
    {
        std::vector<int32_t> ms(nq * o.top);
        std::vector<uint32_t> mi(nq * o.top);
        check(oswald_hip_topr(ctx, (uint32_t)o.top, ms.data(), mi.data()), "top scores");
        for (uint64_t i = 0; i < nq; ++i)
            for (uint64_t j = 0; j < o.top; ++j) { cand_s[i * 2 * o.top + j] = ms[i * o.top + j]; cand_i[i * 2 * o.top + j] = mi[i * o.top + j]; }
        const uint64_t h0 = host_first * W, h1 = db.sequences_count;
        if (h0 < h1)
            for (uint64_t i = 0; i < nq; ++i) {
                std::vector<int32_t> hs;
                std::vector<uint64_t> hi;
                oswald::top_scores(scores.data() + i * row + h0, h1 - h0, o.top, hs, hi);
                for (uint64_t j = 0; j < hs.size(); ++j) { cand_s[i * 2 * o.top + o.top + j] = hs[j]; cand_i[i * 2 * o.top + o.top + j] = (uint32_t)(hi[j] + h0); }
            }
    }
    std::vector<int32_t> fs(nq * o.top);
    std::vector<uint32_t> fi(nq * o.top);
    check(oswald_hip_merge_candidates((uint32_t)nq, 2 * o.top, cand_s.data(), cand_i.data(), (uint32_t)o.top, fs.data(), fi.data()), "merge");
    const double work_time = dwalltime() - tick;
    if (phases) fprintf(stderr, "[oswald] hybrid: test %.1f ms (accelerator %.1f, host %.1f), rest %.1f ms\n", std::max(test_gpu_time, test_cpu_time) * 1e3, test_gpu_time * 1e3, test_cpu_time * 1e3, work_time * 1e3);
    oswald_hip_finalize(ctx);
    std::vector<std::vector<int32_t>> top_s(nq);
    std::vector<std::vector<uint64_t>> top_i(nq);
    for (uint64_t i = 0; i < nq; ++i)
        for (uint64_t j = 0; j < o.top && fs[i * o.top + j] >= 0; ++j) { top_s[i].push_back(fs[i * o.top + j]); top_i[i].push_back(fi[i * o.top + j]); }
    print_report(o, q, db, top_s, top_i, current_time, work_time, work_time + std::max(test_gpu_time, test_cpu_time));
    return 0;
}

// What one device is given in one round of the accelerator-only search: a run of whole groups in the layout the C ABI
// takes, and where its sequences sit in the database.
struct Piece {
    const uint8_t *b = nullptr;
    uint64_t bytes = 0;
    const uint16_t *n = nullptr;
    const uint32_t *disp = nullptr;
    uint32_t ngroups = 0, first_index = 0, nvalid = 0;
    std::vector<uint32_t> index_map;          // dealt pieces: database index of every sequence (not one contiguous run)
    // dealt pieces: their groups copied together, in page-locked memory from the library (oswald_hip_host_alloc: the
    // upload of a piece is then plain asynchronous DMA and its call returns at once -- with N devices driven by one
    // thread, N pageable uploads would be staged one after the other before the last device gets its search)
    struct PinnedBytes {
        uint8_t *p = nullptr;
        PinnedBytes() = default;
        PinnedBytes(const PinnedBytes &) = delete;
        PinnedBytes &operator=(const PinnedBytes &) = delete;
        PinnedBytes(PinnedBytes &&o) noexcept : p(o.p) { o.p = nullptr; }
        PinnedBytes &operator=(PinnedBytes &&o) noexcept { if (this != &o) { release(); p = o.p; o.p = nullptr; } return *this; }
        ~PinnedBytes() { release(); }
        void release() { if (p) (void)oswald_hip_host_free(p); p = nullptr; }
    } owned_b;
    std::vector<uint16_t> owned_n;
    std::vector<uint32_t> owned_disp;
};

// One device: the chunks of the database in order (reference FPGAsearch.c:132-138 with one FPGA).
std::vector<std::vector<Piece>> contiguous_pieces(const oswald::Database &db)
{
    const uint64_t W = oswald::kFpgaVectorLength;
    std::vector<std::vector<Piece>> per_dev(1);
    for (const oswald::Chunk &c : db.chunks) {
        Piece p;
        p.b = c.b; p.bytes = c.b_size; p.n = c.n.data(); p.disp = c.disp.data(); p.ngroups = (uint32_t)c.n.size();
        const uint64_t first = c.accum * W, last = std::min<uint64_t>(db.sequences_count, (c.accum + c.n.size()) * W);
        p.first_index = (uint32_t)first;
        p.nvalid = (uint32_t)(last - first);
        per_dev[0].push_back(std::move(p));
    }
    return per_dev;
}

// Several devices: the length-sorted database is DEALT to them in units of 8 consecutive groups (128 sequences, the
// block one wave of the search kernels works on), counted from the longest end, forwards in even rounds and backwards
// in odd ones -- every device gets the same length distribution and they finish together (1 M sequences over 8 GPUs:
// predicted 7.95 x).  The reference cuts contiguous shards of equal padded size instead (sequences.c:510-515, chunk c
// to device c mod ndev, FPGAsearch.c:132-138): its last shard holds all the longest sequences and takes 1.3 x as long
// as the mean (6.54 x).  A device's units, in ascending order, are copied together into chunks of at most
// max_chunk_size bytes; every chunk carries the database indices of its sequences.  Part of the group assembly: before
// the clock, like assemble_multiple_chunks_db in the reference.
std::vector<std::vector<Piece>> dealt_pieces(const oswald::Database &db, unsigned ndev, uint64_t max_chunk_size, int threads)
{
    const uint64_t W = oswald::kFpgaVectorLength, G = db.vect_sequences_count, N = db.sequences_count, UNIT = 8;
    struct GroupRef { const uint8_t *b; uint16_t n; };
    std::vector<GroupRef> groups;
    groups.reserve(G);
    for (const oswald::Chunk &c : db.chunks)
        for (size_t g = 0; g < c.n.size(); ++g) groups.push_back({c.b + c.disp[g], c.n[g]});
    const uint64_t nunits = (G + UNIT - 1) / UNIT;
    std::vector<std::vector<uint64_t>> units(ndev); // first group of every unit of a device, ascending
    for (uint64_t u = 0; u < nunits; ++u) {
        const uint64_t t = u / ndev, j = u % ndev, d = (t & 1) ? ndev - 1 - j : j;
        units[d].push_back(G > (u + 1) * UNIT ? G - (u + 1) * UNIT : 0);
    }
    std::vector<std::vector<Piece>> per_dev(ndev);
    for (unsigned d = 0; d < ndev; ++d) {
        std::reverse(units[d].begin(), units[d].end());
        auto unit_end = [&](uint64_t g0) { return std::min<uint64_t>(G, g0 == 0 && G % UNIT ? G % UNIT : g0 + UNIT); };
        std::vector<uint64_t> gl;        // the device's groups, ascending
        std::vector<uint8_t> unit_start; // ... and which of them open a unit (chunks are cut there when they can be)
        uint64_t total = 0;
        for (uint64_t g0 : units[d])
            for (uint64_t g = g0; g < unit_end(g0); ++g) { gl.push_back(g); unit_start.push_back(g == g0); total += (uint64_t)groups[g].n * W; }
        const uint64_t parts = std::max<uint64_t>(1, (total + max_chunk_size - 1) / max_chunk_size), target = (total + parts - 1) / parts;
        size_t k = 0;
        while (k < gl.size()) {
            Piece p;
            uint64_t bytes = 0;
            const size_t k0 = k;
            while (k < gl.size()) {
                const uint64_t gb = (uint64_t)groups[gl[k]].n * W;
                if (gb > max_chunk_size) throw std::runtime_error("OSWALD: max_chunk_size is smaller than one group of sequences.");
                if (k > k0 && (bytes + gb > max_chunk_size || (bytes >= target && unit_start[k]))) break;
                bytes += gb;
                ++k;
            }
            { void *mem = nullptr; check(oswald_hip_host_alloc(bytes, &mem), "page-locked piece buffer"); p.owned_b.p = (uint8_t *)mem; }
            for (size_t i = k0; i < k; ++i) {
                const uint64_t g = gl[i];
                p.owned_n.push_back(groups[g].n);
                for (uint64_t l = 0; l < W; ++l) if (g * W + l < N) p.index_map.push_back((uint32_t)(g * W + l));
            }
            p.owned_disp.resize(p.owned_n.size());
            uint64_t off = 0;
            for (size_t g = 0; g < p.owned_n.size(); ++g) { p.owned_disp[g] = (uint32_t)off; off += (uint64_t)p.owned_n[g] * W; }
            p.ngroups = (uint32_t)p.owned_n.size();
            p.nvalid = (uint32_t)p.index_map.size();
            p.bytes = bytes;
            // copy the groups together (the units are contiguous runs of the mapped cache)
#pragma omp parallel for num_threads(threads > 0 ? threads : 1) schedule(static)
            for (long g = 0; g < (long)p.ngroups; ++g) memcpy(p.owned_b.p + p.owned_disp[g], groups[gl[k0 + g]].b, (size_t)p.owned_n[g] * W);
            per_dev[d].push_back(std::move(p));
        }
    }
    for (auto &v : per_dev) for (Piece &p : v) { p.b = p.owned_b.p; p.n = p.owned_n.data(); p.disp = p.owned_disp.data(); }
    return per_dev;
}

int do_search(Options &o)
{
    if (o.execution_mode == 2) return do_search_host_only(o);
    if (o.execution_mode == 1) return do_search_hybrid(o);
    const time_t current_time = time(nullptr);
    printf("\nOSWALD v%s \n\n", oswald::kVersion);
    printf("Database file:\t\t\t%s\n", o.db);
    const bool phases = getenv("OSWALD_DEBUG_PHASES") != nullptr; // wall time of the host-side phases, to stderr
    double tp = dwalltime();
    auto lap = [&](const char *what) { if (phases) { const double t = dwalltime(); fprintf(stderr, "[oswald] %-34s %8.3f ms\n", what, (t - tp) * 1e3); tp = t; } };
    oswald::Queries q = oswald::load_query_sequences(o.queries);
    lap("load queries");
    const uint64_t nq = q.m.size(), W = oswald::kFpgaVectorLength;

    // device bring-up is outside the timed region, like init() in the reference (main.c:46 vs FPGAsearch.c:80) -- and,
    // like there, it adapts the maximum chunk size to the device's memory before the database is cut (utils.c:162-168)
    oswald_hip_ctx *ctx = nullptr;
    check(bring_up(o, &ctx), "device bring-up");
    lap("device bring-up");
    clamp_chunk_size(o, ctx, nq);
    lap("chunk size limit");

    // The report needs the top-r scores per query only.  For r <= 1024 they are selected on the devices, chunk by
    // chunk, gathered over RCCL and folded on GPU 0 with the reference's tie rule (oswald_hip_topr; utils.c:3-86: equal
    // scores -> later database index first); the full score table (the reference downloads and sorts it,
    // FPGAsearch.c:232, :312-321) is only brought to the host for larger r, by the reference's chunk rule.
    const bool device_top = o.top <= 1024;
    oswald::Database db = oswald::assemble_multiple_chunks_db(o.db, oswald::kFpgaVectorLength, o.max_chunk_size, device_top ? 1 : o.num_devices);
    lap("load database + assemble chunks");
    PinnedResidues pinned(db);
    lap("page-lock the residues");
    print_header(o, db);
    if (db.sequences_count < o.top) o.top = db.sequences_count;
    std::vector<std::vector<Piece>> pieces;
    if (device_top) {
        // (a device's first piece is cut into a head and the rest by the library, at its upload: oswald_hip_chunk_upload_async)
        pieces = o.num_devices > 1 ? dealt_pieces(db, o.num_devices, o.max_chunk_size, (int)std::min<unsigned>((unsigned)std::max(o.cpu_threads, 1), oswald::usable_cpus())) : contiguous_pieces(db);
        lap("deal blocks to the devices");
    }
    std::vector<int32_t> scores;
    if (!device_top) scores.assign(nq * db.vect_sequences_count * W, 0);
    std::vector<std::vector<int32_t>> tmp(o.num_devices);
    // (the device's own working memory -- the kernels' spill scratch -- exists since bring-up, like the FPGA kernel's on-chip buffers since
    // init(); the buffers the reference creates INSIDE its clock -- queries, chunk arrays, profiles, scores: FPGAsearch.c:80, :85-96 --
    // are made inside ours: reserve_slots below)
    // (... and the HOST side of the chunk slots -- page-locked staging -- before it, like the reference's posix_memalign, FPGAsearch.c:69-74)
    auto reserve_slots = [&](bool host_only) {
        if (!device_top) return;
        for (unsigned d = 0; d < pieces.size(); ++d) { // the largest piece of the device, one slot more than it keeps in flight
            uint64_t mb = 0;
            uint32_t mg = 0;
            for (const Piece &p : pieces[d]) { mb = std::max(mb, p.bytes); mg = std::max(mg, p.ngroups); }
            const uint32_t slots = (uint32_t)std::min<size_t>(pieces[d].size() + 1, 4);
            if (mb && host_only) check(oswald_hip_reserve_host(ctx, (int)d, mg, (uint32_t)W, (uint32_t)nq, slots), "host buffers");
            else if (mb) check(oswald_hip_reserve_chunks(ctx, (int)d, mb, mg, (uint32_t)W, (uint32_t)nq, slots), "device buffers");
        }
    };
    reserve_slots(true);
    lap("host buffers (page-locked staging)");

    // OSWALD_DEBUG_REPEAT=n (with OSWALD_DEBUG_PHASES; measurement hook): the timed region is run n times; the report is the FIRST
    // pass's -- what a user gets --, the later passes (same process: clocks, caches, the runtime's queues warm) go to stderr
    const int passes = phases && getenv("OSWALD_DEBUG_REPEAT") ? std::max(1, atoi(getenv("OSWALD_DEBUG_REPEAT"))) : 1;
    double workTime = 0;
    std::vector<std::vector<int32_t>> top_s(nq);
    std::vector<std::vector<uint64_t>> top_i(nq);
    for (int pass = 0; pass < passes; ++pass) {
    if (pass > 0) { check(oswald_hip_wait(ctx, -1), "wait"); check(oswald_hip_release_chunks(ctx, -1), "buffer release"); for (auto &v : top_s) v.clear(); for (auto &v : top_i) v.clear(); tp = dwalltime(); } // (a later pass creates its buffers again, like a new run)
    const double tick = dwalltime();
    check(oswald_hip_set_scoring(ctx, oswald::submat_by_name(o.submat), o.open_gap, o.extend_gap, 0), "scoring setup");
    check(oswald_hip_set_queries(ctx, q.a.data(), q.Q, q.m.data(), q.a_disp.data(), (uint32_t)nq), "query upload");
    lap("  scoring + queries");
    reserve_slots(false);
    lap("  device buffers (chunk slots)");
    if (device_top) {
        check(oswald_hip_topr_begin(ctx, (uint32_t)o.top), "top scores");
        // Round k = piece k of every device.  The uploads of a round are queued on all devices (they overlap), every
        // device is given its search -- the piece's top list is selected and folded on the device behind it --, and
        // the uploads of the rounds to come are queued ahead: they come in over the devices' upload streams while round k
        // is being searched (the reference uploads and searches in turn, FPGAsearch.c:180-223).
        size_t rounds = 0;
        for (const auto &v : pieces) rounds = std::max(rounds, v.size());
        // (two rounds ahead: the copies of round k+2 run beside the search of round k, its re-tile -- a kernel, for which the
        // persistent search grid leaves no room -- when that search drains; the host meanwhile plans and queues round k+1,
        // so the device goes from one search into the next without waiting for the host)
        std::vector<std::vector<int>> h(4, std::vector<int>(o.num_devices, -1));
        auto upload = [&](size_t k) {
            std::vector<int> &hk = h[k % 4];
            // one thread per device (the C ABI allows concurrent calls for DIFFERENT devices of a context): with N devices
            // driven one after the other, device N-1 would get its search N-1 plans late (~1 ms each)
#pragma omp parallel for num_threads((int)pieces.size()) schedule(static, 1) if (pieces.size() > 1)
            for (unsigned d = 0; d < pieces.size(); ++d) {
                hk[d] = -1;
                if (k >= pieces[d].size()) continue;
                const Piece &p = pieces[d][k];
                check(oswald_hip_chunk_upload_async(ctx, (int)d, p.b, p.bytes, p.n, p.disp, p.ngroups, (uint32_t)W, &hk[d]), "chunk upload");
                check(oswald_hip_chunk_set_index(ctx, (int)d, hk[d], p.first_index, p.nvalid, p.index_map.empty() ? nullptr : p.index_map.data()), "chunk index");
            }
            lap("  queue uploads of a round");
        };
        // The first search is queued as soon as its own upload is (the library cuts a large first piece in two, so that the device
        // starts on its head while the rest is still on the link).  Behind the search of round k the uploads of rounds k+1 and
        // k+2 are queued: from page-locked memory an upload call returns at once, and the copy stream runs ahead of the searches.
        size_t next_up = 0;
        upload(next_up++);
        for (size_t k = 0; k < rounds; ++k) {
            std::vector<int> &cur = h[k % 4];
#pragma omp parallel for num_threads((int)pieces.size()) schedule(static, 1) if (pieces.size() > 1)
            for (unsigned d = 0; d < pieces.size(); ++d)
                if (cur[d] >= 0) check(oswald_hip_chunk_search(ctx, (int)d, cur[d], nullptr), "chunk search"); // (plans and queues the search behind the piece's upload; waits for nothing)
            lap("  queue searches of a round");
            const size_t ahead = k + 2;
            while (next_up <= ahead && next_up < rounds) upload(next_up++);
            // the slot goes back LAST: the call returns when the piece's upload has landed (its re-tile runs when the search before
            // it drains), and the uploads of the rounds to come must be on the link by then, not behind that wait
#pragma omp parallel for num_threads((int)pieces.size()) schedule(static, 1) if (pieces.size() > 1)
            for (unsigned d = 0; d < pieces.size(); ++d)
                if (cur[d] >= 0) check(oswald_hip_chunk_release(ctx, (int)d, cur[d]), "chunk release");
            lap("  slots of the round given back");
        }
        // top lists of all queries (inside the timed region: they stand for the download of the score table)
        std::vector<int32_t> ms(nq * o.top);
        std::vector<uint32_t> mi(nq * o.top);
        check(oswald_hip_topr(ctx, (uint32_t)o.top, ms.data(), mi.data()), "top scores");
        lap("  gather the top lists (waits for the devices)");
        for (uint64_t i = 0; i < nq; ++i)
            for (uint64_t j = 0; j < o.top && ms[i * o.top + j] >= 0; ++j) { top_s[i].push_back(ms[i * o.top + j]); top_i[i].push_back(mi[i * o.top + j]); }
    } else {
        // chunk c of a round goes to device c mod ndev (reference FPGAsearch.c:132-138)
        for (size_t k = 0; k < db.chunks.size(); k += o.num_devices) {
            const size_t active = std::min<size_t>(o.num_devices, db.chunks.size() - k);
            for (size_t d = 0; d < active; ++d) {
                const oswald::Chunk &c = db.chunks[k + d];
                tmp[d].resize(nq * c.n.size() * W);
                check(oswald_hip_search_chunk_async(ctx, (int)d, c.b, c.b_size, c.n.data(), c.disp.data(), (uint32_t)c.n.size(),
                                                    (uint32_t)W, tmp[d].data()), "chunk search");
            }
            check(oswald_hip_wait(ctx, -1), "wait");
            for (size_t d = 0; d < active; ++d) {
                const oswald::Chunk &c = db.chunks[k + d];
                const size_t row = c.n.size() * W;
                for (uint64_t qi = 0; qi < nq; ++qi)
                    memcpy(scores.data() + (qi * db.vect_sequences_count + c.accum) * W, tmp[d].data() + qi * row, row * sizeof(int32_t));
            }
        }
    }
    const double pass_time = dwalltime() - tick;
    if (pass == 0) workTime = pass_time;
    if (phases) { fprintf(stderr, "[oswald] %-34s %8.3f ms%s\n", "search (timed region), total", pass_time * 1e3, pass ? "   (a later pass of the same process: OSWALD_DEBUG_REPEAT)" : ""); tp = dwalltime(); }
    }
    oswald_hip_finalize(ctx);
    lap("device release");

    // then only the titles the report prints (the reference loads every title, sequences.c:1096-1127; a
    // 1 M-sequence database has 1 M of them for 10 lines per query)
    if (!device_top)
        for (uint64_t i = 0; i < nq; ++i) oswald::top_scores(scores.data() + i * db.vect_sequences_count * W, db.sequences_count, o.top, top_s[i], top_i[i]);
    lap("top scores");
    print_report(o, q, db, top_s, top_i, current_time, workTime, workTime);
    lap("headers + report");
    return 0;
}

} // namespace

int main(int argc, char *argv[])
{
    // the host's worker threads sleep between their parallel regions instead of spinning: with as many of them as the box has
    // hardware threads (-c 128) the spinning kept the threads that drive the accelerators off the cores (hybrid mode: the
    // accelerator's 10 ms test took 100)
    setenv("OMP_WAIT_POLICY", "PASSIVE", 0);
    static struct argp_option options[] = {
        {0, 0, 0, 0, "OSWALD execution", 1},
        {0, 'O', "<string>", 0, "'preprocess' for database preprocessing, 'search' for database search, 'info' for GPU information [REQUIRED]", 1},
        {0, 0, 0, 0, "preprocess", 2},
        {"input", 'i', "<string>", 0, "Input sequence filename (must be in FASTA format). [REQUIRED]", 2},
        {"output", 'o', "<string>", 0, "Output filename. [REQUIRED]", 2},
        {0, 0, 0, 0, "search", 3},
        {"query", 'q', "<string>", 0, "Input query sequence filename (must be in FASTA format). [REQUIRED]", 3},
        {"db", 'd', "<string>", 0, "Preprocessed database output filename. [REQUIRED]", 3},
        {"sm", 's', "<string>", 0, "Substitution matrix. Supported values: blosum45, blosum50, blosum62, blosum80, blosum90, pam30, pam70, pam250 (default: blosum62).", 3},
        {"gap_open", 'g', "<integer>", 0, "Gap open penalty (default: 10).", 3},
        {"gap_extend", 'e', "<integer>", 0, "Gap extend penalty (default: 2).", 3},
        {"execution_mode", 'm', "<integer>", 0, "0 for accelerator mode, 1 for hybrid mode (host + accelerator), 2 or host-only for host mode (default: 1).", 3},
        {"cpu_threads", 'c', "<integer>", 0, "Number of CPU threads (default: 4).", 3},
        {"vector_length", 'v', "<integer>", 0, "Vector length in host: 16 (SSE4.1 kernel) or 32 (AVX2 kernel) (default: 16).", 3},
        {"cpu_block_width", 'b', "<integer>", 0, "CPU block width (default: 256): the 8-bit stage of the host kernel works through a query in blocks of that many rows (0: the whole query at once).  The reference blocks the database sequence by it; this kernel streams the database column by column and blocks the other axis.  Scores do not depend on it.", 3},
        {"num_fpgas", 'f', "<integer>", 0, "Number of GPUs (the reference's number of FPGAs) (default: 1).", 3},
        {"max_chunk_size", 'k', "<integer>", 0, "Maximum chunk size on the accelerator (bytes, default: 134217728).", 3},
        {"db_percentage", 'p', "<integer>", 0, "Database percentage for testing computational power (hybrid mode only) (default: 0.01).  An upper bound in this build: the host's test is called off once the GPU's is over, and the host is rated on what it finished; the ratings then only size the pieces -- host and GPU take work from the two ends of the database until they meet.", 3},
        {"top", 'r', "<integer>", 0, "Number of scores to show (default: 10).", 3},
        {0}};
    Options o;
    o.arg_count = argc;
    struct argp argp = {options, parse_opt, 0, argp_doc};
    argp_parse(&argp, argc, argv, 0, 0, &o);
    try {
        if (!strcmp(o.op, "preprocess")) return do_preprocess(o);
        if (!strcmp(o.op, "info")) return do_info();
        return do_search(o);
    } catch (const std::exception &e) {
        // file errors print and exit(2)/(3) in the reference (sequences.c:18, :133, :1110)
        printf("%s\n", e.what());
        return 2;
    }
}
