"""How many non-MFMA vector instructions does every MFMA loop of the library carry?  (diagnostic, not product)

In a loop of fp32 MFMAs every vector instruction is MFMA time (~4.8 cycles each on gfx950: the fp32 MFMA executes on the
SIMD's vector ALUs) — and address arithmetic the compiler inserts is vector instructions: ds_read2_b32
reaches 255 dwords from its base; a static LDS offset beyond that, or a runtime index, makes the compiler rebuild a base
per read.  This script compiles every translation unit to ISA (device only) and prints, per basic block with >= 8 MFMAs:
MFMAs, other vector instructions (and per MFMA), scalar instructions, LDS instructions.  Round 4 found k_bwd_wgrad_c32
(2.2 per MFMA) and k_lif_step_c32 (0.7) this way; the hand-tuned sequence kernels sit at 0.0 - 0.15.

usage: python experiments/valu_per_mfma.py [substring of the kernel name]"""


def main():
    import glob
    import os
    import re
    import subprocess
    import sys
    import tempfile

    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    CSRC = os.path.join(ROOT, "snn_modulation_classification_amd", "csrc")
    want = sys.argv[1] if len(sys.argv) > 1 else ""
    tmp = tempfile.mkdtemp()
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
        flags = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-std=c++17", "-S", "--cuda-device-only"]
        if src.endswith("dcll_seq_tiled.hip"):
            flags.append("-fno-slp-vectorize")
        subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["-o", out, src], check=True, stderr=subprocess.DEVNULL)
        kernels, name = {}, None
        for line in open(out):
            m = re.match(r"^(_Z\w+):", line)
            if m:
                name = m.group(1)
                kernels[name] = []
            elif name is not None:
                kernels[name].append(line)
                if "s_endpgm" in line:
                    name = None
        for k, lines in kernels.items():
            dem = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
            if want not in dem:
                continue
            blocks, cur = [], []
            for l in lines:
                if re.match(r"^\.LBB", l):
                    blocks.append(cur)
                    cur = []
                cur.append(l)
            blocks.append(cur)
            for b in blocks:
                m = sum("v_mfma" in l for l in b)
                if m < 8:
                    continue
                v = sum(1 for l in b if re.match(r"\s+v_(?!mfma)", l) and "v_accvgpr" not in l)
                sal = sum(1 for l in b if re.match(r"\s+s_(?!waitcnt|nop)", l))
                ds = sum(1 for l in b if re.match(r"\s+ds_", l))
                print("%-90s mfma %4d  valu %4d (%.2f per mfma)  salu %4d  ds %4d" % (dem[:90], m, v, v / m, sal, ds))


if __name__ == "__main__":
    main()
