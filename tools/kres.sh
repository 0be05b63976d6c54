#!/bin/bash
# usage: tools/kres.sh file.hip  -> per-kernel VGPR / spill / scratch / LDS summary (asm kept in /tmp/t)
f=$(realpath $1)
mkdir -p /tmp/t && cd /tmp/t && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage -save-temps -c $f -o /tmp/t/kres.o 2>&1 | python3 -c '
import sys,re
cur=None; vals={}
for line in sys.stdin:
    if " error" in line: print(line.strip())
    m=re.search(r"Function Name: (\S+)",line)
    if m: cur=m.group(1); vals={}; continue
    m=re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)",line)
    if m and cur:
        vals[m.group(1).strip()]=m.group(2)
        if m.group(1).strip().startswith("LDS Size"):
            print(cur[:95], "VGPR",vals.get("VGPRs"),"vspill",vals.get("VGPRs Spill"),"scratch",vals.get("ScratchSize"),"sspill",vals.get("SGPRs Spill"),"SGPR",vals.get("TotalSGPRs"),"LDS",m.group(2))
'
