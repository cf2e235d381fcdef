// tools/oprate8.hip -- round 4, third session: the sequence-pair cell without its v_perm_b32.
// The sequence-pair cell (one query against the two sequences of a lane: osw_sw_s16 / osw_sw_pk16) pairs the substitution
// scores of the lane's two residues with one v_perm_b32 per row and adds the pair to the diagonal with a packed add: 7.5
// VALU instructions per row where the query-pair cell has 6.5.  v_pk_mad_i16 has a half select PER SOURCE and PER RESULT
// HALF (op_sel / op_sel_hi).  With profile entries of 32 bits per row, {low half: S, high half: 1},
//     v_pk_mad_i16 x, Ea, Eb, D op_sel:[0,1,0] op_sel_hi:[1,0,1]
// computes  x.lo = Ea.lo * Eb.hi + D.lo = S(a) + D.lo  and  x.hi = Ea.hi * Eb.lo + D.hi = S(b) + D.hi : the pairing AND
// the diagonal add in ONE instruction, both entries from the SAME table (6.5 instructions per row).  The price is the
// profile: 128 B per query row instead of 64 (what the query-pair cell has), read as 2-row half-blocks of 32 codes x 8 B
// (= 64 banks: conflict-free ds_read_b64), four ds_read_b64 per 4 rows instead of two.
// This probe (harness of tools/oprate6.hip) (1) checks on the device that the instruction computes exactly that, (2) runs
// the column loop of the cell -- 48-row strip, state in place, three waves per SIMD (168 VGPRs) -- in three forms:
//   perm   the kernel's: 2 ds_read_b64 + 4 v_perm_b32 per 4 rows, loads two blocks ahead
//   mad    4 ds_read_b64 per 4 rows (half-blocks F = rows 0,1 and S = rows 2,3 of a block, each re-loaded for the next block
//          behind the add that consumes it), no v_perm_b32; 1 / 4 / 8 lane groups
//   none   no loads, no perms (the VALU floor of the 6.5-instruction row)
// and reports core-clock cycles per row per SIMD (median / slowest / fastest SIMD of the chip).
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/oprate8 tools/oprate8.hip ; run: tools/oprate8 [columns]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <vector>

struct Stamp { unsigned long long cyc, real, r0, r1; uint32_t hwid, pad; };

#define PA0 "v150"
#define PA1 "v151"
#define PA2 "v152"
#define PA3 "v153"
#define PB0 "v154"
#define PB1 "v155"
#define PB2 "v156"
#define PB3 "v157"
#define PS0 "v148"
#define PS1 "v149"
#define VT "v159"
#define VF "v161"
#define FIXED "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v159", "v161"

#define ROWBODY(DN, X, E, SCMAX)                                        \
    "v_pk_maximum3_f16 " DN ", " X ", " E ", " VF "\n\t"                \
    "v_subrev_u32 " VT ", %[go_], " DN "\n\t" SCMAX                     \
    "v_pk_maximum3_f16 " E ", " E ", " VT ", %[fl_]\n\t"                \
    "v_pk_maximum3_f16 " VF ", " VF ", " VT ", %[fl_]\n\t"              \
    "v_subrev_u32 " VF ", %[ge_], " VF "\n\t"
#define ROWT(XN, DN, SN, X, E, HOOK, SCMAX) "v_pk_add_i16 " XN ", " DN ", " SN " clamp\n\t" HOOK ROWBODY(DN, X, E, SCMAX)
#define ROWM(XN, DN, SA, SB, X, E, HOOK, SCMAX) "v_pk_mad_i16 " XN ", " SA ", " SB ", " DN " op_sel:[0,1,0] op_sel_hi:[1,0,1]\n\t" HOOK ROWBODY(DN, X, E, SCMAX)
#define SCRUN(A, B) "v_pk_maximum3_f16 %[sc_], %[sc_], " A ", " B "\n\t"
#define PERM_A(PS, HI, LO) "v_perm_b32 " PS ", " HI ", " LO ", %[sela_]\n\t"
#define PERM_B(PS, HI, LO) "v_perm_b32 " PS ", " HI ", " LO ", %[selb_]\n\t"

#define LD_B64(LO, HI) "ds_read_b64 " LO ", %[a0_] offset:%[off_]\n\tds_read_b64 " HI ", %[a1_] offset:%[off_]\n\ts_waitcnt lgkmcnt(2)\n\t"
#define BLK_PERM(C1, C3, CLO, CHI, N0, N2)                                                                       \
    ROWT("%[xb_]", "%[D1_]", PS0, "%[x_]", "%[E0_]", PERM_A(PS1, C3, C1), "")                                    \
    ROWT("%[x_]", "%[D2_]", PS1, "%[xb_]", "%[E1_]", PERM_B(PS0, C3, C1) LD_B64(CLO, CHI), SCRUN("%[D1_]", "%[D2_]")) \
    ROWT("%[xb_]", "%[D3_]", PS0, "%[x_]", "%[E2_]", PERM_A(PS1, N2, N0), "")                                    \
    ROWT("%[x_]", "%[D4_]", PS1, "%[xb_]", "%[E3_]", PERM_B(PS0, N2, N0), SCRUN("%[D3_]", "%[D4_]"))

// mad form: F = {v150, v151} rows (0, 1) of the lane's first sequence, {v152, v153} of its second; S = {v154, v155} / {v156, v157}
// rows (2, 3).  Row 0 issues the add of row 1 (the last read of F) and re-loads F for the next block; row 2 that of row 3 (the
// last read of S) and re-loads S; each then waits for the half-block the NEXT add reads (two loads stay in flight).
#define LD_F "ds_read_b64 v[150:151], %[a0_] offset:%[of_]\n\tds_read_b64 v[152:153], %[a1_] offset:%[of_]\n\ts_waitcnt lgkmcnt(2)\n\t"
#define LD_S "ds_read_b64 v[154:155], %[a0_] offset:%[os_]\n\tds_read_b64 v[156:157], %[a1_] offset:%[os_]\n\ts_waitcnt lgkmcnt(2)\n\t"
#define BLK_MAD                                                                                                  \
    ROWM("%[xb_]", "%[D1_]", PA1, PA3, "%[x_]", "%[E0_]", LD_F, "")                                              \
    ROWM("%[x_]", "%[D2_]", PB0, PB2, "%[xb_]", "%[E1_]", "", SCRUN("%[D1_]", "%[D2_]"))                         \
    ROWM("%[xb_]", "%[D3_]", PB1, PB3, "%[x_]", "%[E2_]", LD_S, "")                                              \
    ROWM("%[x_]", "%[D4_]", PA0, PA2, "%[xb_]", "%[E3_]", "", SCRUN("%[D3_]", "%[D4_]"))

// mad128 form: 16-byte entries (4 rows x {S, 1} of one code), one ds_read_b128 per sequence and block; buffer X = v[150:153] (first
// sequence) + v[154:157] (second), buffer Y = v[140:143] + v[144:147]; the blocks alternate, a buffer is re-loaded for the block after
// next behind the add of row 3 (its last read), then the wait for the next block's (two loads stay in flight)
#define LD_128(A, B) "ds_read_b128 " A ", %[a0_] offset:%[o128_]\n\tds_read_b128 " B ", %[a1_] offset:%[o128_]\n\ts_waitcnt lgkmcnt(2)\n\t"
#define BLK_MAD128(A1, A2, A3, B1, B2, B3, CA, CB, NA0, NB0)                                                     \
    ROWM("%[xb_]", "%[D1_]", A1, B1, "%[x_]", "%[E0_]", "", "")                                                  \
    ROWM("%[x_]", "%[D2_]", A2, B2, "%[xb_]", "%[E1_]", "", SCRUN("%[D1_]", "%[D2_]"))                           \
    ROWM("%[xb_]", "%[D3_]", A3, B3, "%[x_]", "%[E2_]", LD_128(CA, CB), "")                                      \
    ROWM("%[x_]", "%[D4_]", NA0, NB0, "%[xb_]", "%[E3_]", "", SCRUN("%[D3_]", "%[D4_]"))

#define BLK_NONE                                                                                                 \
    ROWT("%[xb_]", "%[D1_]", "%[s1_]", "%[x_]", "%[E0_]", "", "")                                                \
    ROWT("%[x_]", "%[D2_]", "%[s2_]", "%[xb_]", "%[E1_]", "", SCRUN("%[D1_]", "%[D2_]"))                         \
    ROWT("%[xb_]", "%[D3_]", "%[s1_]", "%[x_]", "%[E2_]", "", "")                                                \
    ROWT("%[x_]", "%[D4_]", "%[s2_]", "%[xb_]", "%[E3_]", "", SCRUN("%[D3_]", "%[D4_]"))

#define STMT(TXT, RB, NEXTOFF, NEXTB)                                                                            \
    asm volatile(TXT                                                                                             \
                 : [x_] "+v"(x), [xb_] "=&v"(xb), [E0_] "+v"(E[RB * 4]), [E1_] "+v"(E[RB * 4 + 1]), [E2_] "+v"(E[RB * 4 + 2]),  \
                   [E3_] "+v"(E[RB * 4 + 3]), [D1_] "+v"(D[RB * 4 + 1]), [D2_] "+v"(D[RB * 4 + 2]), [D3_] "+v"(D[RB * 4 + 3]),  \
                   [D4_] "+v"(D[RB * 4 + 4]), [sc_] "+v"(sc)                                                    \
                 : [a0_] "v"(a0), [a1_] "v"(a1), [off_] "i"(NEXTOFF * 256), [of_] "i"(NEXTB * 512), [os_] "i"(NEXTB * 512 + 256), [o128_] "i"(NEXTOFF * 512),  \
                   [ge_] "s"(ge), [go_] "s"(go), [fl_] "v"(fl),                                                 \
                   [sela_] "s"(0x05040100u), [selb_] "s"(0x07060302u), [s1_] "v"(s1), [s2_] "v"(s2)                \
                 : "memory", FIXED)

enum { V_PERM = 0, V_MAD = 1, V_NONE = 2, V_MAD128 = 3 };

template <int V, int RB>
static __device__ __forceinline__ void block(uint32_t a0, uint32_t a1, uint32_t (&D)[49], uint32_t (&E)[48], uint32_t &x, uint32_t &sc, uint32_t ge,
                                             uint32_t go, uint32_t fl, uint32_t s1, uint32_t s2)
{
    constexpr int NEXT = (RB + 2) % 12, NEXTB = (RB + 1) % 12; // (the last blocks of a column load the first again: the stream never stops)
    uint32_t xb;
    if constexpr (V == V_PERM) {
        if constexpr ((RB & 1) == 0) STMT(BLK_PERM(PA1, PA3, "v[150:151]", "v[152:153]", PB0, PB2), RB, NEXT, NEXTB);
        else STMT(BLK_PERM(PB1, PB3, "v[154:155]", "v[156:157]", PA0, PA2), RB, NEXT, NEXTB);
    } else if constexpr (V == V_MAD) {
        STMT(BLK_MAD, RB, NEXT, NEXTB);
    } else if constexpr (V == V_MAD128) {
        if constexpr ((RB & 1) == 0) STMT(BLK_MAD128("v151", "v152", "v153", "v155", "v156", "v157", "v[150:153]", "v[154:157]", "v140", "v144"), RB, NEXT, NEXTB);
        else STMT(BLK_MAD128("v141", "v142", "v143", "v145", "v146", "v147", "v[140:143]", "v[144:147]", "v150", "v154"), RB, NEXT, NEXTB);
    } else {
        STMT(BLK_NONE, RB, NEXT, NEXTB);
    }
    if constexpr (RB + 1 < 12) block<V, RB + 1>(a0, a1, D, E, x, sc, ge, go, fl, s1, s2);
}

#define LDS_WORDS 13312 // 52 KB: eight lane groups x (48 rows x 128 B + 16)

template <int V, int G>
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_vgpr(140))) void probe(Stamp *out, uint32_t seed, int ncols)
{
    extern __shared__ uint32_t lds[];
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // perm: entries of 8 B = 4 rows x int16 of one code; mad: entries of 8 B = 2 rows x {S, 1} of one code
    for (uint32_t i = threadIdx.x; i < LDS_WORDS; i += 256) lds[i] = (V == V_MAD || V == V_MAD128) ? (0x00010000u | ((i * 7 + 3) & 3)) : 0x00010001u * ((i * 7 + 3) & 3);
    __syncthreads();
    constexpr uint32_t SLICE = (V == V_MAD || V == V_MAD128) ? 6144u : 3072u; // bytes of a 48-row slice
    const uint32_t base = G == 1 ? wave * SLICE : (lane / (64 / G)) * (SLICE + ((V == V_MAD || V == V_MAD128) ? 16u : 8u));
    uint32_t D[49], E[48], x = 0x04000400u, sc = 0x04000400u, fl = 0x04000400u, s1 = 0x00010002u, s2 = 0x00020001u;
    const uint32_t ge = 2, go = 10;
#pragma unroll
    for (int i = 0; i < 48; ++i) { D[i] = 0x04000400u + lane; E[i] = 0x04000400u; }
    D[48] = 0x04000400u;
    asm volatile("v_mov_b32 " VF ", %0\n\tv_mov_b32 " PS0 ", 0\n\tv_mov_b32 " PS1 ", 0\n\t"
                 "v_mov_b32 " PA0 ", 0\n\tv_mov_b32 " PA1 ", 0\n\tv_mov_b32 " PA2 ", 0\n\tv_mov_b32 " PA3 ", 0\n\t"
                 "v_mov_b32 " PB0 ", 0\n\tv_mov_b32 " PB1 ", 0\n\tv_mov_b32 " PB2 ", 0\n\tv_mov_b32 " PB3 ", 0" ::"v"(fl) : FIXED);
    uint32_t rng = seed + threadIdx.x * 2654435761u + blockIdx.x * 40503u;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int col = 0; col < ncols; ++col) {
        rng = rng * 1664525u + 1013904223u;
        constexpr uint32_t ES = V == V_MAD128 ? 16u : 8u; // bytes per profile entry
        const uint32_t a0 = base + ((((rng >> 8) & 0xffu) * 20u) >> 8) * ES, a1 = base + ((((rng >> 20) & 0xffu) * 20u) >> 8) * ES;
        if constexpr (V == V_PERM) {
            asm volatile("ds_read_b64 v[150:151], %[a0_]\n\tds_read_b64 v[152:153], %[a1_]\n\t"
                         "ds_read_b64 v[154:155], %[a0_] offset:256\n\tds_read_b64 v[156:157], %[a1_] offset:256\n\ts_waitcnt lgkmcnt(2)\n\t" PERM_A(PS1, PA2, PA0)
                             PERM_B(PS0, PA2, PA0) "v_pk_add_i16 %[x_], %[tp_], " PS1 " clamp"
                         : [x_] "=&v"(x)
                         : [a0_] "v"(a0), [a1_] "v"(a1), [tp_] "v"(D[0]), [sela_] "s"(0x05040100u), [selb_] "s"(0x07060302u)
                         : "memory", FIXED);
        } else if constexpr (V == V_MAD) {
            asm volatile("ds_read_b64 v[150:151], %[a0_]\n\tds_read_b64 v[152:153], %[a1_]\n\t"
                         "ds_read_b64 v[154:155], %[a0_] offset:256\n\tds_read_b64 v[156:157], %[a1_] offset:256\n\ts_waitcnt lgkmcnt(2)\n\t"
                         "v_pk_mad_i16 %[x_], " PA0 ", " PA2 ", %[tp_] op_sel:[0,1,0] op_sel_hi:[1,0,1]"
                         : [x_] "=&v"(x)
                         : [a0_] "v"(a0), [a1_] "v"(a1), [tp_] "v"(D[0])
                         : "memory", FIXED);
        } else if constexpr (V == V_MAD128) {
            asm volatile("ds_read_b128 v[150:153], %[a0_]\n\tds_read_b128 v[154:157], %[a1_]\n\t"
                         "ds_read_b128 v[140:143], %[a0_] offset:512\n\tds_read_b128 v[144:147], %[a1_] offset:512\n\ts_waitcnt lgkmcnt(2)\n\t"
                         "v_pk_mad_i16 %[x_], v150, v154, %[tp_] op_sel:[0,1,0] op_sel_hi:[1,0,1]"
                         : [x_] "=&v"(x)
                         : [a0_] "v"(a0), [a1_] "v"(a1), [tp_] "v"(D[0])
                         : "memory", FIXED);
        } else {
            asm volatile("v_pk_add_i16 %[x_], %[tp_], %[s1_] clamp" : [x_] "=&v"(x) : [tp_] "v"(D[0]), [s1_] "v"(s1));
        }
        block<V, 0>(a0, a1, D, E, x, sc, ge, go, fl, s1, s2);
        D[0] = D[48];
        if constexpr (V != V_NONE) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory", FIXED); // (the wrapped loads of the last blocks)
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    uint32_t acc = sc ^ x;
#pragma unroll
    for (int i = 0; i < 48; ++i) acc ^= D[i] ^ E[i];
    if (lane == 0) {
        Stamp s;
        s.cyc = t1 - t0 + (acc == 0x12345678u);
        s.real = r1 - r0;
        s.r0 = r0;
        s.r1 = r1;
        s.hwid = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (15 << 11));
        s.pad = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        out[blockIdx.x * 4 + wave] = s;
    }
}

// what the instruction computes: out[i] = v_pk_mad_i16(ea[i], eb[i], d[i]) with the cell's half selects
__global__ void mad_check(const uint32_t *ea, const uint32_t *eb, const uint32_t *d, uint32_t *out, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t x;
    asm volatile("v_pk_mad_i16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(x) : "v"(ea[i]), "v"(eb[i]), "v"(d[i]));
    out[i] = x;
}

struct Probe { const char *name; void (*kern)(Stamp *, uint32_t, int); const char *what; };

int main(int argc, char **argv)
{
    hipDeviceProp_t p;
    (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    {   // (1) the instruction: every score -128..127 for both sequences against diagonal values over the cells' whole range
        const int n = 256 * 256 * 4;
        std::vector<uint32_t> ea(n), eb(n), d(n), got(n);
        const uint32_t dv[4] = {0x04000400u, 0x7b707b70u, 0x04007b70u, 0x56781234u};
        for (int i = 0; i < n; ++i) {
            const int sa = (i & 255) - 128, sb = ((i >> 8) & 255) - 128;
            ea[i] = 0x00010000u | (uint16_t)sa;
            eb[i] = 0x00010000u | (uint16_t)sb;
            d[i] = dv[i >> 16];
        }
        uint32_t *da, *db, *dd, *dout;
        (void)hipMalloc(&da, n * 4); (void)hipMalloc(&db, n * 4); (void)hipMalloc(&dd, n * 4); (void)hipMalloc(&dout, n * 4);
        (void)hipMemcpy(da, ea.data(), n * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(db, eb.data(), n * 4, hipMemcpyHostToDevice);
        (void)hipMemcpy(dd, d.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mad_check, dim3(n / 256), dim3(256), 0, 0, da, db, dd, dout, n);
        (void)hipMemcpy(got.data(), dout, n * 4, hipMemcpyDeviceToHost);
        long bad = 0;
        for (int i = 0; i < n; ++i) {
            const int sa = (i & 255) - 128, sb = ((i >> 8) & 255) - 128;
            const uint32_t want = (uint32_t)(uint16_t)((d[i] & 0xffffu) + sa) | ((uint32_t)(uint16_t)((d[i] >> 16) + sb) << 16);
            if (got[i] != want && bad++ < 5) printf("   MISMATCH sa %d sb %d d %08x: got %08x want %08x\n", sa, sb, d[i], got[i], want);
        }
        printf("v_pk_mad_i16 x, {S(a),1}, {S(b),1}, D op_sel:[0,1,0] op_sel_hi:[1,0,1] == {D.lo + S(a), D.hi + S(b)} on %d cases: %ld mismatches\n", n, bad);
        (void)hipFree(da); (void)hipFree(db); (void)hipFree(dd); (void)hipFree(dout);
    }
    const Probe probes[] = {
        {"perm", probe<V_PERM, 1>, "2 ds_read_b64 + 4 v_perm_b32 per 4 rows (the kernel's sequence-pair cell, 7.5 VALU per row)"},
        {"perm8", probe<V_PERM, 8>, "... 8 lane groups on one shared slice"},
        {"mad", probe<V_MAD, 1>, "4 ds_read_b64 per 4 rows, v_pk_mad_i16 pairs and adds (6.5 VALU per row), 32-bit {S, 1} profile entries"},
        {"mad4", probe<V_MAD, 4>, "... 4 lane groups on one shared slice"},
        {"mad8", probe<V_MAD, 8>, "... 8 lane groups"},
        {"m128", probe<V_MAD128, 1>, "2 ds_read_b128 per 4 rows (16-byte entries: 4 rows x {S, 1}; 16 buffer registers), v_pk_mad_i16 (6.5 VALU per row)"},
        {"m128x4", probe<V_MAD128, 4>, "... 4 lane groups on one shared slice"},
        {"m128x8", probe<V_MAD128, 8>, "... 8 lane groups"},
        {"none", probe<V_NONE, 1>, "no loads, no perms (6.5 VALU per row)"},
    };
    const int ncols = argc > 1 ? atoi(argv[1]) : 4000;
    Stamp *o;
    (void)hipMalloc(&o, (size_t)cus * 8 * 4 * sizeof(Stamp));
    printf("device %s, %d CUs; core-clock cycles per ROW (48-row columns, %d columns) per SIMD, over the SIMDs of the chip\n", p.gcnArchName, cus, ncols);
    const int wpss[] = {1, 2, 3};
    for (const Probe &pr : probes) {
        printf("%-5s %s\n", pr.name, pr.what);
        for (int wps : wpss) {
            const int nb = cus * wps;
            const size_t lds = (size_t)(160 * 1024 / wps) - 1024;
            (void)hipFuncSetAttribute((const void *)pr.kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            hipLaunchKernelGGL(pr.kern, dim3(nb), dim3(256), lds, 0, o, 12345u, ncols);
            if (hipDeviceSynchronize() != hipSuccess) { printf("   launch failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
            std::vector<Stamp> h((size_t)nb * 4);
            (void)hipMemcpy(h.data(), o, h.size() * sizeof(Stamp), hipMemcpyDeviceToHost);
            struct Slot { unsigned long long r0 = ~0ull, r1 = 0; int n = 0; double clk = 0; };
            std::map<uint32_t, Slot> slots;
            double clk = 0;
            for (const Stamp &s : h) {
                Slot &sl = slots[((s.hwid >> 4) & 3u) | (((s.hwid >> 8) & 0xffu) << 2) | (s.pad << 10)];
                sl.r0 = std::min(sl.r0, s.r0);
                sl.r1 = std::max(sl.r1, s.r1);
                sl.n++;
                sl.clk += (double)s.cyc / ((double)s.real / 100.0);
                clk += (double)s.cyc / ((double)s.real / 100.0);
            }
            std::vector<double> cpr;
            int most = 0;
            for (auto &kv : slots) {
                const Slot &sl = kv.second;
                most = std::max(most, sl.n);
                const double cycles = (double)(sl.r1 - sl.r0) / 100.0 * (sl.clk / sl.n);
                cpr.push_back(cycles / ((double)ncols * 48.0 * sl.n));
            }
            std::sort(cpr.begin(), cpr.end());
            printf("   w%d: median %6.2f  slowest %6.2f  fastest %6.2f  @%4.0f MHz  (%zu SIMDs, at most %d waves on one)\n", wps, cpr[cpr.size() / 2],
                   cpr.back(), cpr.front(), clk / h.size(), slots.size(), most);
        }
        fflush(stdout);
    }
    return 0;
}
