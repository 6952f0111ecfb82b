"""Per-basic-block instruction table of one kernel (diagnostic, not product).

valu_per_mfma.py looks only at the blocks that hold MFMAs; a loop is more than that block.  This prints EVERY basic block of a
kernel — MFMAs, other vector instructions, scalar instructions, LDS and memory instructions, its branches, and (with -v) the
vector opcodes — so that address arithmetic (v_add_co / v_addc / v_lshl_add_u64 / v_mad_u64_u32), exec-mask branches around
conditional loads and speculated code show up wherever in the loop they sit.  Round 4 found the 43 address instructions and
~10 branches per chunk of k_lif_step_c32 and the 21 + 6 per K-chunk of k_readout_t16 with it.

usage: python experiments/isa_blocks.py <file.hip under csrc> <substring of the MANGLED kernel name> [-v]
  e.g. python experiments/isa_blocks.py dcll_hip.hip k_lif_step_c32ILb1ELi0ELb0 -v"""


def main():
    import os
    import re
    import subprocess
    import sys
    import tempfile

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = os.path.join(ROOT, "snn_modulation_classification_amd", "csrc", sys.argv[1])
    pat, verbose = sys.argv[2], "-v" in sys.argv[3:]
    out = os.path.join(tempfile.mkdtemp(), "k.s")
    flags = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-S", "--cuda-device-only"]
    if src.endswith("dcll_seq_tiled.hip"):
        flags.append("-fno-slp-vectorize")
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-o", out, src], check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and pat in l]
    if not starts:
        sys.exit("no kernel matches %r" % pat)
    start = starts[0]
    end = next(k for k in range(start, len(lines)) if "s_endpgm" in lines[k])
    print(subprocess.run(["c++filt", lines[start].split(":")[0]], capture_output=True, text=True).stdout.strip()[:150])
    blocks = []
    for l in lines[start:end + 1]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            blocks.append(dict(name=m.group(1), mfma=0, valu=0, salu=0, ds=0, mem=0, br=[], ops={}))
            continue
        t = l.strip().split(" ")[0] if l.strip() else ""
        if not blocks or not t or t[0] in ";.":
            continue
        b = blocks[-1]
        if t.startswith("v_mfma"):
            b["mfma"] += 1
        elif t.startswith("v_"):
            b["valu"] += 1
            b["ops"][t] = b["ops"].get(t, 0) + 1
        elif t.startswith("s_"):
            b["salu"] += 1
            if t.startswith("s_cbranch") or t == "s_branch":
                b["br"].append(l.strip().replace("s_cbranch_", "").replace(".LBB", ""))
        elif t.startswith("ds_"):
            b["ds"] += 1
        elif t.startswith(("global_", "buffer_", "flat_", "scratch_")):
            b["mem"] += 1
    for b in blocks:
        print("%-12s mfma %4d  valu %4d  salu %4d  ds %4d  mem %4d  %s  %s" %
              (b["name"], b["mfma"], b["valu"], b["salu"], b["ds"], b["mem"], b["br"],
               dict(sorted(b["ops"].items(), key=lambda kv: -kv[1])) if verbose and b["valu"] > 8 else ""))


if __name__ == "__main__":
    main()
