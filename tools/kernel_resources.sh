# prints VGPRs / spills / scratch / LDS / occupancy of the kernels in zdr_kernels.hip whose name matches $1 (default: k_path)
# (hipcc -Rpass-analysis=kernel-resource-usage; honours ZDR_KERNEL_FLAGS like zdr_amd/build.py)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -O3 -munsafe-fp-atomics -fno-slp-vectorize $ZDR_KERNEL_FLAGS -Iinclude -Izdr_amd/csrc \
  -Rpass-analysis=kernel-resource-usage -c zdr_amd/csrc/zdr_kernels.hip -o /tmp/zdr_kernels_res.o 2>&1 | grep "remark:" | python3 -c "
import sys, re, subprocess
pat = sys.argv[1]
cur = None; rows = []
for line in sys.stdin:
    t = line.split('remark:', 1)[1].replace('[-Rpass-analysis=kernel-resource-usage]', '').strip()
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':', 1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1); cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(['c++filt', r['name']], capture_output=True, text=True).stdout.strip().split('(')[0].replace('void ', '')
    if pat not in name: continue
    print(f\"{name:58s} VGPR {r.get('VGPRs','?'):>4s} spill {r.get('VGPRs Spill','?'):>4s} scratch {r.get('ScratchSize [bytes/lane]','?'):>5s} LDS {r.get('LDS Size [bytes/block]','?'):>6s} occ {r.get('Occupancy [waves/SIMD]','?')}\")
" "${1:-k_path}"
