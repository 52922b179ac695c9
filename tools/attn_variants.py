"""Experiment builds of the attention kernels WITHOUT touching cosa_amd/csrc (the teacher kernels' source hash is part of the accuracy record):
a patched copy of attn_kernels.hip is compiled (both operand builds) and linked with the tree's other objects into
cosa_amd/lib/variants/libcosa_hip_<name>.so.  usage: python tools/attn_variants.py <name> [<name> ...]; A/B: tools/ab_attn_libs.py"""
import os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cosa_amd import build as B

SRC = open(os.path.join(B.CSRC, "attn_kernels.hip")).read()
a0, a1 = SRC.index("template <bool DMA, int OUTM = 0, int NW = 2, bool AUGM = false>"), SRC.index("// backward (student pass)")
FWD2 = SRC[a0:a1]


CMAX = False          # replace the inline-asm v_max3_f32 helper by fmaxf chains the compiler sees (its hazard recognizer covers them)


def fwd2(f):
    out = SRC[:a0] + f(FWD2) + SRC[a1:]
    if CMAX:
        i = out.index('    float r;\n    asm("v_max3_f32 %0, %1, %2, %3"')
        j = out.index("    return r;\n", i) + len("    return r;\n")
        out = out[:i] + "    return __builtin_fmaxf(__builtin_fmaxf(a, b), c);\n" + out[j:]
    return out


def noprio(t):
    return re.sub(r"^\s*__builtin_amdgcn_s_setprio\(\d\);\n", "", t, flags=re.M)


def prio_pv_only(t):          # priority only around the PV MFMAs
    i = t.index("__builtin_amdgcn_s_setprio(1);")
    j = t.index("__builtin_amdgcn_s_setprio(0);", i)
    t = t[:i] + t[i:j + 40].replace("__builtin_amdgcn_s_setprio(1);", "").replace("__builtin_amdgcn_s_setprio(0);", "") + t[j + 40:]
    return t


def prio_qk_only(t):
    i = t.rindex("__builtin_amdgcn_s_setprio(1);")
    j = t.index("__builtin_amdgcn_s_setprio(0);", i)
    return t[:i] + t[i:j + 40].replace("__builtin_amdgcn_s_setprio(1);", "").replace("__builtin_amdgcn_s_setprio(0);", "") + t[j + 40:]


def occ3(t):                  # three workgroups of 256 per CU (<= 168 VGPRs)
    return t.replace("__launch_bounds__(256, 2) void attn_fwd2_kernel", "__launch_bounds__(256, 3) void attn_fwd2_kernel")



def _pipe(t, sgb):
    """two query blocks of a wave software-pipelined: QK(u1) issues inside softmax(u0)'s exponentials, PV(u0) inside softmax(u1)'s; the same
    operations in the same order per accumulator chain -> bit-identical"""
    a = t.index("        op16x2 pk[2][2][8];")
    b = t.index("        if (DMA) {                                     // the next tile has landed")
    old = t[a:b]
    # the softmax of one query block = body of `for (int u = 0; u < 2; u++) {` ... up to the PV section
    i0 = old.index("        for (int u = 0; u < 2; u++) {\n            if constexpr (tail) {")
    pv = "#pragma unroll\n        for (int kb = 0; kb < 2; kb++)\n#pragma unroll\n            for (int sp = 0; sp < 2; sp++) {"
    i1 = old.index(pv)
    if old[:i1].endswith("        __builtin_amdgcn_s_setprio(1);\n"):
        i1 -= len("        __builtin_amdgcn_s_setprio(1);\n")
    body = old[i0:i1]
    body = body[body.index("{\n") + 2:]
    body = body[:body.rindex("        }\n")]            # drop the loop's closing brace
    # split at the probabilities: [mask + max + rescale branch] | [exp + row sums + packing]
    k = body.index("            // probabilities: exp2, rounded to the operand type in pairs")
    part_a, part_b = body[:k], body[k:]
    M = "COSA_MFMA_32x32x16"
    new = f"""        op16x2 pk[2][2][8];
        f32x16 sc[2][2];
        op16x8 ka[4][2];
#pragma unroll
        for (int s = 0; s < 4; s++) {{
            const int slot = ((2 * s + hh) ^ swz) << 4;
            ka[s][0] = *reinterpret_cast<const op16x8 *>(Ks + r * 128 + slot);
            ka[s][1] = *reinterpret_cast<const op16x8 *>(Ks + (r + 32) * 128 + slot);
        }}
        auto qk = [&](const int u) {{
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int i = 0; i < 16; i++) sc[u][kb][i] = 0.f;
            if (AUGM) {{
                sc[u][0] = {M}(a_aug, qaug[u], sc[u][0], 0, 0, 0);
                sc[u][1] = {M}(a_aug, qaug[u], sc[u][1], 0, 0, 0);
            }}
#pragma unroll
            for (int s = 0; s < 4; s++) {{
                sc[u][0] = {M}(ka[s][0], qf[u][s], sc[u][0], 0, 0, 0);
                sc[u][1] = {M}(ka[s][1], qf[u][s], sc[u][1], 0, 0, 0);
            }}
        }};
        auto smax_a = [&](const int u) {{
{part_a}        }};
        auto smax_b = [&](const int u) {{
{part_b}        }};
        auto pv = [&](const int u) {{
#pragma unroll
            for (int kb = 0; kb < 2; kb++)
#pragma unroll
                for (int sp = 0; sp < 2; sp++) {{
                    const int keyb = kb * 32 + 16 * sp + 4 * hh;
                    const op16x8 v0 = v_frag(Vs, keyb, 0, lane), v1 = v_frag(Vs, keyb, 1, lane);
                    const op16x8 pf = __builtin_shufflevector(__builtin_shufflevector(pk[u][kb][4 * sp], pk[u][kb][4 * sp + 1], 0, 1, 2, 3),
                                                              __builtin_shufflevector(pk[u][kb][4 * sp + 2], pk[u][kb][4 * sp + 3], 0, 1, 2, 3),
                                                              0, 1, 2, 3, 4, 5, 6, 7);
                    o[u][0] = {M}(v0, pf, o[u][0], 0, 0, 0);
                    o[u][1] = {M}(v1, pf, o[u][1], 0, 0, 0);
                }}
        }};
        qk(0);
        smax_a(0);
        smax_b(0);
        qk(1);
{sgb[0]}        smax_a(1);
        smax_b(1);
        pv(0);
{sgb[1]}        pv(1);
        }}
"""
    return t[:a] + new + t[b:]


def v_pipe(t):
    return noprio(_pipe(t, ("", "")))


def _sgb(n_mfma, n_valu):
    return "".join(f"        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);\n        __builtin_amdgcn_sched_group_barrier(0x002, {n_valu}, 0);\n" for _ in range(n_mfma))


def v_pipe_sgb(t):
    return noprio(_pipe(t, (_sgb(10, 8), _sgb(8, 10))))



def bk2(t):
    """NW = 4 (long sequences): K / V staged two 64-key tiles at a time (ring of 2 x 32 KB): one barrier + vmcnt(0) per 128 keys instead of per
    64; the compute walks the same 64-key tiles in the same order -> bit-identical.  NW = 2 keeps the 32-KB ring (four workgroups per CU)."""
    def rep(a, b):
        nonlocal t
        assert t.count(a) == 1, (t.count(a), a)
        t = t.replace(a, b)
    rep("    __shared__ __attribute__((aligned(16))) unsigned char smem[(DMA ? 4 : 2) * BK * 128];",
        "    constexpr bool SUP = DMA && NW == 4;\n    __shared__ __attribute__((aligned(16))) unsigned char smem[(SUP ? 8 : (DMA ? 4 : 2)) * BK * 128];")
    rep("""        dma_tile(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");""", """        dma_tile(0, 0);
        if (SUP && BK < N) dma_tile(BK, 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");""")
    rep("    auto tile = [&](int k0, auto tail_tag) {", "    auto tile = [&](int k0, auto tail_tag, const int sub, const bool sync_after) {")
    rep("""            Ks = smem + ring * 2 * BK * 128;
            Vs = Ks + BK * 128;
            if (k0 + BK < N) dma_tile(k0 + BK, ring ^ 1);""", """            if (SUP) {
                Ks = smem + (ring * 2 + sub) * 2 * BK * 128;
                if (sub == 0) {
                    if (k0 + 2 * BK < N) dma_tile(k0 + 2 * BK, (ring ^ 1) * 2);
                    if (k0 + 3 * BK < N) dma_tile(k0 + 3 * BK, (ring ^ 1) * 2 + 1);
                }
            } else {
                Ks = smem + ring * 2 * BK * 128;
                if (k0 + BK < N) dma_tile(k0 + BK, ring ^ 1);
            }
            Vs = Ks + BK * 128;""")
    rep("        if (DMA) {                                     // the next tile has landed and nobody still reads this one", "        if (DMA && sync_after) {                       // the next tile(s) landed and nobody still reads this one")
    rep("""    for (int k0 = 0; k0 < nfull; k0 += BK) tile(k0, std::false_type{});
    if (nfull < N) tile(nfull, std::true_type{});""", """    int tcount = 0;
    for (int k0 = 0; k0 < nfull; k0 += BK, tcount++) tile(k0, std::false_type{}, SUP ? (tcount & 1) : 0, !SUP || (tcount & 1) || k0 + BK >= N);
    if (nfull < N) tile(nfull, std::true_type{}, SUP ? (tcount & 1) : 0, true);""")
    return t



def regstage(t):
    """K / V tiles through registers instead of LDS-DMA (guide T14): the next tile's 16-byte pieces are fetched with plain buffer loads at the
    top of a tile (16 VGPRs at NW = 4) and written to the other ring buffer after the PV MFMAs; the LDS image is the same -> bit-identical"""
    def rep(a, b):
        nonlocal t
        assert t.count(a) == 1, (t.count(a), a)
        t = t.replace(a, b)
    rep("""    auto dma_tile = [&](int k0, int buf) {""", """    u32x4v stK[4], stV[4];
    auto fetch_tile = [&](int k0) {
        const unsigned ko = (unsigned)((size_t)k0 * rs * 2);
#pragma unroll
        for (int i = 0; i < PW; i++) {
            stK[i] = __builtin_amdgcn_raw_buffer_load_b128(rsK, voK[i] + ko, 0, 0);
            stV[i] = __builtin_amdgcn_raw_buffer_load_b128(rsV, voV[i] + ko, 0, 0);
        }
    };
    auto commit_tile = [&](int buf) {
        unsigned char *kd = smem + buf * 2 * BK * 128 + (PW * wave) * 1024 + lane * 16;
#pragma unroll
        for (int i = 0; i < PW; i++) {
            *reinterpret_cast<u32x4v *>(kd + i * 1024) = stK[i];
            *reinterpret_cast<u32x4v *>(kd + BK * 128 + i * 1024) = stV[i];
        }
    };
    auto dma_tile = [&](int k0, int buf) {""")
    rep("""        dma_tile(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();""", """        fetch_tile(0);
        commit_tile(0);
        __syncthreads();""")
    rep("""            if (k0 + BK < N) dma_tile(k0 + BK, ring ^ 1);""", """            if (k0 + BK < N) fetch_tile(k0 + BK);""")
    rep("""        if (DMA) {                                     // the next tile has landed and nobody still reads this one
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();""", """        if (DMA) {                                     // the next tile: registers -> the other buffer (nobody reads it during this tile)
            if (k0 + BK < N) commit_tile(ring ^ 1);
            __syncthreads();""")
    return "typedef unsigned u32x4v __attribute__((ext_vector_type(4)));\n" + t



def occ1(t):
    """one workgroup per CU (LDS padded to 96 KB): how much do the two co-resident waves of a SIMD hide of each other?"""
    a = "    __shared__ __attribute__((aligned(16))) unsigned char smem[(DMA ? 4 : 2) * BK * 128];"
    assert t.count(a) == 1
    return t.replace(a, "    __shared__ __attribute__((aligned(16))) unsigned char smem[(DMA ? 4 : 2) * BK * 128 + (NW == 4 ? 64 * 1024 : 0)];")



def timing(t):
    """s_memtime stamps at the section boundaries of a tile, summed per wave and added up in stamps[128 ..] (tools/ab_attn_libs.py prints them with
    ATTN_TIMING=1): where do a wave's cycles go?  (the stamps wait for outstanding LDS reads: the sections are slightly serialised)"""
    def rep(a, b, cnt=1):
        nonlocal t
        assert t.count(a) == cnt, (t.count(a), a)
        t = t.replace(a, b)
    rep("    int ring = 0;\n", "    int ring = 0;\n    unsigned long long tsum[6] = {0, 0, 0, 0, 0, 0}, tprev = __builtin_readcyclecounter();\n"
        "#define TS(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); tsum[i] += t_ - tprev; tprev = t_; } while (0)\n")
    rep("        if (!DMA) __syncthreads();\n#pragma unroll\n        for (int i = 0; i < (DMA ? 0 : PW); i++) {", "        TS(0);\n        if (!DMA) __syncthreads();\n#pragma unroll\n        for (int i = 0; i < (DMA ? 0 : PW); i++) {")
    # after the QK MFMAs: right before the softmax loop over u
    rep("#pragma unroll\n        for (int u = 0; u < 2; u++) {\n            if constexpr (tail) {", "        TS(1);\n#pragma unroll\n        for (int u = 0; u < 2; u++) {\n            if constexpr (tail) {")
    # before the PV section
    rep("#pragma unroll\n        for (int kb = 0; kb < 2; kb++)\n#pragma unroll\n            for (int sp = 0; sp < 2; sp++) {", "        TS(2);\n#pragma unroll\n        for (int kb = 0; kb < 2; kb++)\n#pragma unroll\n            for (int sp = 0; sp < 2; sp++) {")
    rep("        if (DMA) {                                     // the next tile has landed and nobody still reads this one\n", "        TS(3);\n        if (DMA) {                                     // the next tile has landed and nobody still reads this one\n")
    rep("            ring ^= 1;\n        }\n    };\n", "            ring ^= 1;\n        }\n        TS(4);\n        tsum[5] += 1;\n    };\n")
    rep("    if (stamps) {\n        __syncthreads();", "    if (stamps && lane == 0 && busy) {\n        for (int i = 0; i < 6; i++) atomicAdd(&stamps[128 + i], tsum[i]);\n    }\n    if (stamps) {\n        __syncthreads();")
    return t



def vprefetch(t, early=True):
    """PV section with the V^T fragments of key group g + 1 requested BEFORE the four MFMAs of group g are issued (the compiler's own schedule
    waits for each group's transposing reads right in front of its MFMAs: ~10 exposed LDS latencies per tile), the first group's reads above the
    softmax; the same MFMAs in the same order -> bit-identical"""
    a = t.index("#pragma unroll\n        for (int kb = 0; kb < 2; kb++)\n#pragma unroll\n            for (int sp = 0; sp < 2; sp++) {\n                const int keyb = kb * 32 + 16 * sp + 4 * hh;")
    b = t.index("        if (DMA) {                                     // the next tile has landed", a)
    new_pv = """        {
            op16x8 vc0 = vpre0, vc1 = vpre1;
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const int kb = g >> 1, sp = g & 1;
                op16x8 vn0 = vc0, vn1 = vc1;
                if (g < 3) {
                    const int keyn = ((g + 1) >> 1) * 32 + 16 * ((g + 1) & 1) + 4 * hh;
                    vn0 = v_frag(Vs, keyn, 0, lane);
                    vn1 = v_frag(Vs, keyn, 1, lane);
                }
#pragma unroll
                for (int u = 0; u < 2; u++) {
                    const op16x8 pf = __builtin_shufflevector(__builtin_shufflevector(pk[u][kb][4 * sp], pk[u][kb][4 * sp + 1], 0, 1, 2, 3),
                                                              __builtin_shufflevector(pk[u][kb][4 * sp + 2], pk[u][kb][4 * sp + 3], 0, 1, 2, 3),
                                                              0, 1, 2, 3, 4, 5, 6, 7);
                    o[u][0] = COSA_MFMA_32x32x16(vc0, pf, o[u][0], 0, 0, 0);
                    o[u][1] = COSA_MFMA_32x32x16(vc1, pf, o[u][1], 0, 0, 0);
                }
                vc0 = vn0; vc1 = vn1;
            }
        }
        }
"""
    t = t[:a] + new_pv + t[b:]
    # the first group's fragments: requested before the softmax (V does not depend on it)
    anchor = "#pragma unroll\n        for (int u = 0; u < 2; u++) {\n            if constexpr (tail) {"
    assert t.count(anchor) == 1
    t = t.replace(anchor, "        const op16x8 vpre0 = v_frag(Vs, 4 * hh, 0, lane), vpre1 = v_frag(Vs, 4 * hh, 1, lane);\n" + anchor)
    return t


VARIANTS = {"base": lambda t: t, "noprio": noprio, "prio_pv": prio_pv_only, "prio_qk": prio_qk_only, "occ3": occ3,
            "occ3_noprio": lambda t: occ3(noprio(t)), "regstage": regstage, "occ1": occ1, "vprefetch": vprefetch, "occ1_vprefetch": lambda t: occ1(vprefetch(t)), "timing": timing, "occ1_timing": lambda t: occ1(timing(t)), "occ1_pipe": lambda t: occ1(v_pipe(t)), "occ1_regstage": lambda t: occ1(regstage(t)), "bk2": lambda t: bk2(noprio(t)), "bk2_pipe": lambda t: bk2(v_pipe(t))}
VARIANTS.update({k: v for k, v in globals().items() if k.startswith("v_") and callable(v)})


FLAG_VARIANTS = {"bias0": ["-mllvm", "-amdgpu-schedule-metric-bias=0"],
                 "iter_ilp": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"],
                 "iter_minreg": ["-mllvm", "-amdgpu-sched-strategy=iterative-minreg"],
                 "max_ilp": ["-mllvm", "-amdgpu-sched-strategy=max-ilp"]}          # compiler-flag variants of the unpatched source


def build(name):
    global CMAX
    CMAX = name.endswith("_cmax")
    top = os.path.join("/tmp/attnv", name)
    shutil.rmtree(top, ignore_errors=True)
    d = os.path.join(top, "cosa_amd", "csrc")
    shutil.copytree(B.CSRC, d)
    os.symlink(os.path.join(ROOT, "include"), os.path.join(top, "include"))          # (common.hpp includes ../../include/cosa_hip.h)
    flags = FLAG_VARIANTS.get(name, [])
    open(os.path.join(d, "attn_kernels.hip"), "w").write(fwd2(VARIANTS["base" if name in FLAG_VARIANTS else (name[:-5] if CMAX else name)]))
    hipcc = B._hipcc()
    objs = []
    for key, extra in B.SOURCES.items():
        src, _, tag = key.partition("@")
        op = os.path.join(B.OBJDIR, src.replace(".hip", ("_" + tag if tag else "") + ".o"))
        if src == "attn_kernels.hip":
            op = os.path.join(d, "attn" + ("_" + tag if tag else "") + ".o")
            subprocess.check_call([hipcc, "-c", os.path.join(d, src), "-o", op] + B.COMMON + extra + flags)
        objs.append(op)
    out = os.path.join(ROOT, "cosa_amd", "lib", "variants")
    os.makedirs(out, exist_ok=True)
    lib = os.path.join(out, f"libcosa_hip_{name}.so")
    subprocess.check_call([hipcc, "-shared", "-fPIC", "--offload-arch=" + B.ARCH, "-o", lib] + objs)
    print(lib)


if __name__ == "__main__":
    for n in sys.argv[1:]:
        build(n)
