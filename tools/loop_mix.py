#!/usr/bin/env python3
"""Instruction mix of the meshlet cull kernel's main loop, from the gfx950 assembly (no GPU needed).

  python tools/loop_mix.py [--flags 7] [--table 1] [-D TR_...]        -> prints a summary (profiles/r3/loop_mix_*.txt)

Compiles toyrenderer_amd/csrc/k_basepass_as.hip to device assembly with the product's flags, takes
meshletCullKernel<FRUSTUM, OCCLUSION, CONE, TABLE>, finds the innermost loop that holds the LDS-DMA prefetches and walks
its HOT path: straight-line from the loop header, conditional branches not taken (they lead to the exact-arithmetic path,
the deferred-lookup note and the zero-weight check: all out of line and rare), `s_cbranch_execz` over the table lookup
not taken (the lookup executes), unconditional branches followed, until the header is reached again.  One trip round the
loop = kRingSlots steps of 64 meshlets; the counts are printed per step.
"""
import argparse
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "toyrenderer_amd", "csrc", "k_basepass_as.hip")
FLAGS = "-std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fno-slp-vectorize -Wno-unused-function -Wno-inline-asm --offload-arch=gfx950 --cuda-device-only -S".split()


def classify(m: str) -> str:
    if m.startswith("v_pk_"):
        return "VALU packed fp32" if m.endswith("_f32") else "VALU packed other"
    if m in ("v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_exp_f32", "v_log_f32", "v_rcp_iflag_f32"):
        return "VALU transcendental"
    if m.startswith("v_cmp") or m.startswith("v_cmpx"):
        return "VALU compare"
    if m.startswith("v_cndmask"):
        return "VALU select"
    if m.startswith("v_readlane") or m.startswith("v_readfirstlane") or m.startswith("v_writelane"):
        return "VALU lane access"
    if m.startswith("v_"):
        if re.search(r"_f32(_e\d+)?$", m) or m in ("v_fmac_f32", "v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32"):
            return "VALU fp32"
        return "VALU integer / move / convert"
    if m.startswith("ds_"):
        return "LDS"
    if m.startswith("global_load_lds"):
        return "VMEM LDS-DMA"
    if m.startswith("global_") or m.startswith("buffer_") or m.startswith("flat_"):
        return "VMEM"
    if m == "s_nop":
        return "s_nop"
    if m == "s_waitcnt":
        return "s_waitcnt"
    if m.startswith("s_cbranch") or m == "s_branch":
        return "SALU branch"
    if m.startswith("s_load") or m.startswith("s_buffer_load"):
        return "SMEM"
    if m.startswith("s_"):
        return "SALU"
    return "other"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flags", type=int, default=7)
    ap.add_argument("--table", type=int, default=1)
    ap.add_argument("-D", action="append", default=[])
    ap.add_argument("--asm", help="use this assembly file instead of compiling")
    ap.add_argument("--dump", action="store_true", help="print the traced instructions")
    ap.add_argument("--mode", default="cont", choices=["cont", "alone"], help="which instantiation of the body: continuous mode (default) or stand-alone batches")
    args = ap.parse_args()
    if args.asm:
        text = open(args.asm).read()
    else:
        with tempfile.TemporaryDirectory() as td:
            out = os.path.join(td, "k.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + ["-D" + d for d in args.D] + [SRC, "-o", out], stderr=subprocess.DEVNULL)
            text = open(out).read()
    b = lambda x: "Lb1E" if x else "Lb0E"
    sym = "meshletCullKernelI" + b(args.flags & 1) + b(args.flags & 2) + b(args.flags & 4) + b(args.table) + "EEv"
    lines = text.split("\n")
    start = next(i for i, l in enumerate(lines) if sym in l and re.match(r"^_Z\S+:", l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end")) - 1
    # resource lines after the body
    meta = {}
    for l in lines[end:end + 120]:
        m = re.match(r"\s*;\s*(NumVgprs|NumSgprs|Occupancy|LDSByteSize|ScratchSize|NumAgprs|TotalNumVgprs):\s*(\d+)", l)
        if m:
            meta[m.group(1)] = int(m.group(2))
    body = lines[start:end + 1]
    labels = {}
    for i, l in enumerate(body):
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = i
    # innermost loop holding the DMA prefetch: the header label is the last "Parent Loop ... Depth=2"-style header before the
    # first vmcnt wait that follows a label; find loop headers by their comment
    headers = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l) and ("Parent Loop" in l or "This Inner Loop" in l or "=>This Loop Header" in l)]
    dma = [i for i, l in enumerate(body) if "global_load_lds" in l]
    cands = []
    for h in headers:
        if "Parent Loop" not in body[h] and "Inner Loop" not in body[h]:
            continue
        nxt = [d for d in dma if d > h]
        nxt_header = [x for x in headers if x > h]
        if nxt and nxt[0] - h < 600 and (not nxt_header or nxt[0] < nxt_header[0]):
            cands.append(h)
    assert cands, "main loop not found"
    # the kernel body exists twice: batches that stand alone (record order) first, continuous mode (tile-ordered list) last
    cand = cands[0] if args.mode == "alone" else cands[-1]
    header = re.match(r"^(\.LBB\d+_\d+):", body[cand]).group(1)
    counts = collections.Counter()
    mnems = collections.Counter()
    trace = []
    i = cand + 1
    steps = 0
    guard = 0
    while guard < 20000:
        guard += 1
        if i == cand and trace:
            break                  # fell through into the header again: one trip done
        l = body[i].split(";")[0].strip()
        if not l or l.startswith(".") and not l.startswith(".LBB") or re.match(r"^\.LBB\d+_\d+:", l):
            i += 1
            continue
        parts = l.split()
        m = re.sub(r"_e(32|64)$", "", parts[0])
        cat = classify(m)
        counts[cat] += 1
        mnems[m] += 1
        trace.append(l)
        if m == "s_branch":
            tgt = parts[1]
            if tgt == header:
                break
            i = labels[tgt]
            continue
        if m.startswith("s_cbranch"):
            tgt = parts[1]
            if tgt == header:      # the back edge, taken
                break
            i += 1                 # not taken (see docstring)
            continue
        i += 1
    nd = sum(1 for t in trace if t.startswith("global_load_lds"))
    steps = nd // 2
    valu = sum(v for k, v in counts.items() if k.startswith("VALU"))
    print(f"kernel meshletCullKernel<{args.flags & 1},{(args.flags >> 1) & 1},{(args.flags >> 2) & 1},{args.table}>  body: {args.mode}  defines {args.D}")
    print("registers / LDS:", meta)
    print(f"hot path of one trip round the main loop: {len(trace)} instructions = {steps} steps of 64 meshlets")
    print(f"per step: {valu / steps:.1f} VALU, {counts['SALU'] / steps:.1f} SALU (+ {counts['SALU branch'] / steps:.1f} branches), "
          f"{counts['s_nop'] / steps:.1f} s_nop, {counts['LDS'] / steps:.1f} LDS, {(counts['VMEM'] + counts['VMEM LDS-DMA']) / steps:.1f} VMEM, {counts['s_waitcnt'] / steps:.1f} s_waitcnt")
    for k in sorted(counts):
        print(f"   {k:32s} {counts[k] / steps:7.1f}")
    print("most frequent mnemonics per step:")
    for m, c in mnems.most_common(40):
        print(f"   {m:28s} {c / steps:6.1f}")
    if args.dump:
        print("\n".join(trace))


if __name__ == "__main__":
    main()
