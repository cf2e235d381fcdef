#!/usr/bin/env python3
"""tools/oprate5_gen.py > tools/oprate5.hip -- generator of the s_nop placement study (round 4).

The column-frame cell's row (add, max3, sub, max3, max3, sub, + 1/2 max3) as ONE asm statement per four rows, with an
`s_nop 0` behind chosen instructions of every row (positions 1..6; 7 = behind the half max3 of the odd rows), or with two
alternating temporaries.  tools/oprate4.hip showed that the rows cost LESS with the hazard recogniser's s_nop between
them than without (25.9 against 26.7 cycles at three waves per SIMD): this finds where a pause pays."""
import itertools

VARIANTS = [("none", ())] + [("p%d" % i, (i,)) for i in range(1, 8)] + \
           [("p%d%d" % (a, b), (a, b)) for a, b in ((1, 4), (2, 5), (3, 6), (1, 6), (2, 6), (4, 6), (5, 6), (2, 4), (3, 5))] + \
           [("p135", (1, 3, 5)), ("p246", (2, 4, 6)), ("all", (1, 2, 3, 4, 5, 6))] + \
           [("p" + "".join(map(str, c)), c) for k in (2, 3, 4) for c in itertools.combinations((1, 3, 4, 6), k) if c not in ((1, 4), (3, 6), (1, 6), (4, 6))] + \
           [("p13_7", (1, 3, 7)), ("p36_7", (3, 6, 7)), ("p14_7", (1, 4, 7))]
EXTRA = ["even6", "odd6", "t2_none", "t2_p6", "blk"]   # nop behind row (6) of even rows only / odd rows only; two temporaries; one nop per 4 rows


def row(xn, x, E, Dn, t, nops, sc=None):
    n = lambda k: "s_nop 0\\n\\t" if k in nops else ""
    s = f'"v_add_u32 {xn}, {Dn}, %[s]\\n\\t{n(1)}"\n'
    s += f'"v_pk_maximum3_f16 {Dn}, {x}, {E}, %[Fc]\\n\\t{n(2)}"\n'
    s += f'"v_subrev_u32 {t}, %[go], {Dn}\\n\\t{n(3)}"\n'
    s += f'"v_pk_maximum3_f16 {E}, {E}, {t}, %[fl]\\n\\t{n(4)}"\n'
    s += f'"v_pk_maximum3_f16 %[Fc], %[Fc], {t}, %[fl]\\n\\t{n(5)}"\n'
    s += f'"v_subrev_u32 %[Fc], %[ge], %[Fc]\\n\\t{n(6)}"\n'
    if sc:
        s += f'"v_pk_maximum3_f16 %[sc], %[sc], {sc[0]}, {sc[1]}\\n\\t{n(7)}"\n'
    return s


def stmt(e0, d, nops_per_row, two_t):
    rows = ""
    names = [("%[xb]", "%[x]"), ("%[x]", "%[xb]"), ("%[xb]", "%[x]"), ("%[x]", "%[xb]")]
    for r in range(4):
        t = "%[t2]" if (two_t and r % 2) else "%[t]"
        sc = (f"%[D{r}]", f"%[D{r + 1}]") if r % 2 else None
        rows += row(names[r][0], names[r][1], f"%[E{r}]", f"%[D{r + 1}]", t, nops_per_row[r], sc)
    outs = '[x] "+v"(xx), [xb] "=&v"(xn), [t] "=&v"(t), [t2] "=&v"(t2), ' + ", ".join(f'[E{r}] "+v"(E[{e0 + r}])' for r in range(4)) + ", " + \
           ", ".join(f'[D{r + 1}] "+v"(D[{d[r]}])' for r in range(4)) + ', [Fc] "+v"(Fc), [sc] "+v"(sc)'
    return f'asm volatile({rows}: {outs} : [s] "v"(c1), [go] "v"(go), [ge] "v"(ge), [fl] "v"(fl));\n'


def kernel(name, nops_rows, two_t=False):
    body = stmt(0, (1, 2, 3, 4), nops_rows[:4], two_t) + stmt(4, (5, 6, 7, 0), nops_rows[4:], two_t)
    return f'''__global__ __launch_bounds__(256) void p_{name}(Stamp *out, uint32_t c1, uint32_t c2, int iters)
{{
    extern __shared__ uint32_t pad_lds[];
    if (iters < 0) pad_lds[threadIdx.x] = c1;
    uint32_t x[16], D[8], E[8], Fc = c2, xx = c2, sc = c2, xn, t, t2;
    uint32_t go = 0x000a000au, ge = 0x00020002u, fl = c2;
    asm volatile("" : "+v"(go), "+v"(ge), "+v"(fl));
    for (int i = 0; i < 16; ++i) x[i] = 0;
    for (int i = 0; i < 8; ++i) {{ D[i] = c2 + threadIdx.x; E[i] = c2; }}
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {{
{body}    }}
    x[0] = sc ^ xx ^ Fc;
    for (int i = 0; i < 8; ++i) x[1] ^= D[i] ^ E[i];
    PROBE_EPILOGUE
}}
'''


print(open(__file__.replace("oprate5_gen.py", "oprate5_head.inc")).read())
names = []
for name, nops in VARIANTS:
    print(kernel(name, [nops] * 8))
    names.append(name)
print(kernel("even6", [(6,) if r % 2 == 0 else () for r in range(8)])); names.append("even6")
print(kernel("odd6", [(6,) if r % 2 else () for r in range(8)])); names.append("odd6")
print(kernel("odd7", [(7,) if r % 2 else () for r in range(8)])); names.append("odd7")
print(kernel("blk", [(6,) if r % 4 == 3 else () for r in range(8)])); names.append("blk")
print(kernel("t2_none", [()] * 8, True)); names.append("t2_none")
print(kernel("t2_p6", [(6,)] * 8, True)); names.append("t2_p6")
print("struct Probe { const char *name; void (*kern)(Stamp *, uint32_t, uint32_t, int); };")
print("static const Probe probes[] = {" + ", ".join('{"%s", p_%s}' % (n, n) for n in names) + "};")
print(open(__file__.replace("oprate5_gen.py", "oprate5_main.inc")).read())
