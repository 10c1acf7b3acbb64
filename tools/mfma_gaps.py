#!/usr/bin/env python3
"""Static MFMA-gap histogram of a kernel: compiles a .hip source to gfx950 assembly (product flags) and, for every stretch of code
between two s_barrier instructions that contains MFMAs, lists the instructions the compiler placed between consecutive MFMAs --
by class (VALU, transcendental, LDS, SALU, waitcnt / nop, VMEM, branch) -- and the histogram of gap lengths.  A 32x32x16 bf16 MFMA
occupies the matrix pipe for 32 cycles = ~8 issue slots: a gap of more than ~5 other instructions from ONE wave cannot hide behind
it (MI355X_MICROARCH.md, per-instruction cycle constants); with two compute waves per SIMD the partner's MFMAs fill such gaps only
if the partner is in an MFMA phase at that moment.
usage: mfma_gaps.py <file.hip> <kernel-name substring>     (CPU only: needs hipcc, no GPU)"""
import collections, os, re, subprocess, sys, tempfile
src, want = sys.argv[1], sys.argv[2]
inc = os.path.dirname(os.path.abspath(src))
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fvisibility=hidden", "-mllvm", "-amdgpu-mfma-vgpr-form",
                    "-S", "--cuda-device-only", src, "-o", out], check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
m = [k for k in re.findall(r"^(_Z\w+):", text, re.M) if want in k]
assert m, f"no kernel matching {want}"
name = m[0]
body = text[text.index(name + ":"):]
body = body[:body.index("s_endpgm")]
ins = [ln.strip().split()[0] for ln in body.splitlines() if ln.startswith("\t") and ln.strip() and not ln.strip().startswith((";", "."))]


def cls(op):
    if op.startswith("v_mfma"): return "MFMA"
    if op.startswith(("v_exp", "v_rcp", "v_rsq", "v_log", "v_sqrt", "v_sin", "v_cos")): return "trans"
    if op.startswith("v_"): return "VALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith(("s_waitcnt", "s_nop")): return "wait/nop"
    if op.startswith(("s_cbranch", "s_branch", "s_barrier", "s_setprio", "s_sleep")): return "ctl"
    if op.startswith("s_"): return "SALU"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")): return "VMEM"
    return "other"


print(f"kernel {name}: {len(ins)} instructions, {sum(1 for i in ins if cls(i) == 'MFMA')} MFMAs")
seg, segs = [], []
for op in ins:
    if op == "s_barrier":
        segs.append(seg); seg = []
    else:
        seg.append(op)
segs.append(seg)
for si, seg in enumerate(segs):
    n_mfma = sum(1 for o in seg if cls(o) == "MFMA")
    if n_mfma < 4:
        continue
    gaps, cur, started = [], [], False
    for o in seg:
        if cls(o) == "MFMA":
            if started:
                gaps.append(cur)
            cur, started = [], True
        elif started:
            cur.append(o)
    tot = collections.Counter(cls(o) for o in seg)
    hist = collections.Counter(min(len(g), 12) for g in gaps)
    between = collections.Counter(c for g in gaps for c in map(cls, g))
    print(f"-- code between barriers #{si}: {len(seg)} instructions {dict(tot)}")
    print(f"   {n_mfma} MFMAs; other instructions BETWEEN consecutive MFMAs: {sum(len(g) for g in gaps)} {dict(between)}")
    print("   gap length (other instructions between two MFMAs) -> count: " + ", ".join(f"{k if k < 12 else '12+'}: {hist[k]}" for k in sorted(hist)))
    longest = sorted(gaps, key=len, reverse=True)[:2]
    for g in longest:
        if len(g) > 5:
            print(f"   longest gaps: {len(g)} instructions: {dict(collections.Counter(map(cls, g)))}")
