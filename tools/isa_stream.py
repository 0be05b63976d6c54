#!/usr/bin/env python3
"""One line per basic block of a kernel's gfx950 assembly, every instruction as one letter -- to SEE what the scheduler did
with a hand-built phase (are the vector instructions spread over the MFMA gaps or clustered; where do the waits sit):
   python tools/isa_stream.py uz_conv3x3_pp.hip 'ELi8ELi1ELi3EEELb0ELb0ELb1E' [--min=20]
     M v_mfma   v other VALU   D ds_read / ds_write   B buffer / global memory   L LDS-DMA   w s_waitcnt   | s_barrier
     p s_setprio   s other SALU   j branch
The second argument is a substring of the mangled kernel name.  hipcc cross-compiles here (no GPU needed)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "unet_zoo_amd", "csrc")


def cls(t, line):
    if t.startswith("v_mfma"):
        return "M"
    if t.startswith("v_"):
        return "v"
    if t.startswith("ds_"):
        return "D"
    if t.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "L" if " lds" in line else "B"
    if t == "s_barrier":
        return "|"
    if t == "s_waitcnt":
        return "w"
    if t == "s_setprio":
        return "p"
    if t.startswith(("s_cbranch", "s_branch")):
        return "j"
    if t.startswith("s_"):
        return "s"
    return "?"


def main(argv):
    src, pat = argv[0], argv[1]
    minlen = 20
    for a in argv[2:]:
        if a.startswith("--min="):
            minlen = int(a.split("=")[1])
    path = src if os.path.isabs(src) else os.path.join(CSRC, src)
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S",
                        "-Wno-unused-command-line-argument", path, "-o", out] + [a for a in argv[2:] if a.startswith("-D")],
                       check=True, cwd=td)
        text = open(out).read()
    m = re.search(r"^(_Z\S*" + re.escape(pat) + r"\S*):", text, re.M)
    if not m:
        print("no kernel matches", pat)
        return 1
    start = m.end()
    end = text.index(".Lfunc_end", start)
    print("#", m.group(1))
    block, label = "", "entry"
    for line in text[start:end].split("\n"):
        st = line.strip()
        if not st or st.startswith(";") or st.startswith(".") and not st.endswith(":"):
            continue
        if st.endswith(":") or re.match(r"^\.LBB\S+:", st):
            if len(block) >= minlen:
                print(f"{label:12s} {len(block):5d}  {block}")
            block, label = "", st.split(":")[0]
            continue
        t = st.split()[0]
        block += cls(t, st)
    if len(block) >= minlen:
        print(f"{label:12s} {len(block):5d}  {block}")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
