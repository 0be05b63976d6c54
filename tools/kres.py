#!/usr/bin/env python3
"""Per-kernel register / spill / scratch / LDS table of the kernel library, read from the code-object metadata.

    python tools/kres.py                      every .hip of unet_zoo_amd/csrc (four compiles at a time)
    python tools/kres.py uz_gemm_dma.hip ...  the named files
    python tools/kres.py --spills             only kernels with a spill or scratch
    python tools/kres.py --loads              kernels whose loads are waited for one by one (load -> s_waitcnt vmcnt(0) -> load)

hipcc cross-compiles gfx950 without a GPU (`--cuda-device-only -S`); the numbers are the `amdhsa.kernels` entries of the
emitted assembly (`.vgpr_count` includes the accumulator registers, `.vgpr_spill_count`, `.private_segment_fixed_size` =
scratch bytes per lane), i.e. what the loader will allocate -- not the remarks of an analysis pass.
`tests/test_kernel_resources.py` runs `collect()` over the tree and fails on any spill outside its allow-list: round 4
shipped a 105-register spill on the ConvTranspose input-gradient GEMM that one look at this table would have caught.
"""
import concurrent.futures
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "unet_zoo_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only", "-S", "-Wno-unused-command-line-argument"]

_KEYS = ("agpr_count", "vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size",
         "group_segment_fixed_size")


def demangle(names):
    try:
        out = subprocess.run(["c++filt"] + list(names), capture_output=True, text=True, check=True)
        return out.stdout.strip().split("\n")
    except Exception:
        return list(names)


def parse_asm(text):
    """[{name, vgpr_count, ...}] from the amdhsa.kernels list of one .s file."""
    i = text.find("amdhsa.kernels:")
    if i < 0:
        return []
    kernels, cur = [], None
    for line in text[i:].split("\n")[1:]:
        if line.startswith("amdhsa.") or line.startswith("..."):
            break
        if line.startswith("  - "):          # a new kernel entry starts at list level
            cur = {}
            kernels.append(cur)
            line = "    " + line[4:]
        m = re.match(r"^    \.(\w+):\s+(\S+)\s*$", line)
        if m and cur is not None:
            k, v = m.group(1), m.group(2)
            if k in _KEYS:
                cur[k] = int(v)
            elif k == "name":
                cur["name"] = v
    return [k for k in kernels if "name" in k]


def load_wait_points(text):
    """{kernel symbol: (points, loads)}: `points` = places where a global / buffer load is followed within 25 instructions by
    `s_waitcnt vmcnt(0)` with more loads still to come -- the signature of loads that run one memory round trip after the
    other (hipcc 7.2 closes a block that contains a load with that wait, so `if (in range) v = *p;` serialises an
    unrolled group of loads and a prefetch waits for its own data).  LDS-DMA loads (`... lds`) are not counted."""
    lines = text.split("\n")
    out, cur, start = {}, None, 0
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            cur, start = m.group(1), i
        elif l.startswith(".Lfunc_end") and cur is not None:
            body = [x for x in lines[start:i] if x.startswith("\t") and not x.startswith("\t.") and not x.startswith("\t;")]
            ev = []
            for j, x in enumerate(body):
                if re.search(r"\b(global_load|buffer_load|flat_load)", x) and " lds" not in x:
                    ev.append((j, "L"))
                elif re.search(r"s_waitcnt.*vmcnt\(0\)", x):
                    ev.append((j, "W"))
            pts = 0
            for k, (j, t) in enumerate(ev):
                if t == "W" and k > 0 and ev[k - 1][1] == "L" and j - ev[k - 1][0] <= 25 and any(t2 == "L" for _, t2 in ev[k + 1:]):
                    pts += 1
            out[cur] = (pts, sum(1 for _, t in ev if t == "L"))
            cur = None
    return out


def compile_one(path, extra=()):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run([HIPCC] + FLAGS + list(extra) + [path, "-o", out], capture_output=True, text=True, cwd=td)
        if r.returncode != 0:
            raise RuntimeError(f"{path}: hipcc failed\n{r.stderr[-2000:]}")
        with open(out) as f:
            text = f.read()
        ks = parse_asm(text)
        lw = load_wait_points(text)
    for k in ks:
        k["load_wait_points"], k["loads"] = lw.get(k["name"], (0, 0))
    names = demangle([k["name"] for k in ks])
    for k, n in zip(ks, names):
        k["demangled"] = n
        k["file"] = os.path.basename(path)
    return ks


def collect(files=None, jobs=4, extra=()):
    if not files:
        files = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    paths = [f if os.path.isabs(f) else os.path.join(CSRC, os.path.basename(f)) for f in files]
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as ex:
        res = list(ex.map(lambda p: compile_one(p, extra), paths))
    return [k for ks in res for k in ks]


def short(name, n=110):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\((anonymous namespace::)?\w*Args\w*\)$|\(.*\)$", "", name)
    return name if len(name) <= n else name[: n - 3] + "..."


def main(argv):
    only_spills = "--spills" in argv
    files = [a for a in argv if not a.startswith("--")]
    ks = collect(files)
    if "--loads" in argv:     # kernels whose loads are waited for one by one, worst first
        for k in sorted(ks, key=lambda k: -k["load_wait_points"]):
            if k["load_wait_points"] >= 2:
                print(f"{k['load_wait_points']:3d} load -> vmcnt(0) -> load points / {k['loads']:3d} loads  {k['file']:18s} {short(k['demangled'])}")
        return 0
    bad = 0
    for k in ks:
        sp = k.get("vgpr_spill_count", 0) or k.get("private_segment_fixed_size", 0)
        bad += bool(sp)
        if only_spills and not sp:
            continue
        print(f"{k['file']:20s} {short(k['demangled']):110s} vgpr {k.get('vgpr_count', 0):3d} (agpr {k.get('agpr_count', 0):3d}) "
              f"sgpr {k.get('sgpr_count', 0):3d} vspill {k.get('vgpr_spill_count', 0):3d} sspill {k.get('sgpr_spill_count', 0):3d} "
              f"scratch {k.get('private_segment_fixed_size', 0):4d} lds {k.get('group_segment_fixed_size', 0):6d}")
    print(f"# {len(ks)} kernels, {bad} with spills or scratch")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
