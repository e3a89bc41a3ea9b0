# developer tool: registers / scratch / spills per kernel (cross-compiles, no GPU needed)
cd $(dirname $(readlink -f $0))/../phy-engine_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -I../include -std=c++20 -O3 -fPIC -DNDEBUG -c -o /tmp/k.o -x hip pe_kernels.hip -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
name=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: name=m.group(1); d={}; continue
    m=re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\w+) \[-R',l)
    if m and name:
        d[m.group(1)]=m.group(2)
        if m.group(1).startswith('LDS'): print(name[6:40], ' '.join(f'{k}={v}' for k,v in d.items() if k in ('TotalSGPRs','VGPRs','ScratchSize','SGPRs Spill','VGPRs Spill','Occupancy')))
"
