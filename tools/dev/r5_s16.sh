#!/bin/bash
set -u
OUT=gpurun_out/r5p; mkdir -p $OUT
export TMPDIR=/tmp
timeout 200 python bench.py --steps 16 --warmup 4 --no-cpu-baseline --loop-steps 0 --no-profile --option search_stats=1 > $OUT/st.json 2> $OUT/st.err
python - <<'PY'
import re
frames=[]; cur=None
for line in open('gpurun_out/r5p/st.err'):
    if line.startswith('[icp stats]'):
        cur={'stats':line.strip(),'ph':{}}; frames.append(cur)
    m=re.match(r'\[icp phases\] it\s+(\d+): start skew ([\d.]+), span ([\d.]+) us; A mean ([\d.]+) max ([\d.]+); B mean ([\d.]+) max ([\d.]+) \(that block: (\d+) misses\).*misses (\d+) in (\d+) blocks', line)
    if m and cur is not None:
        cur['ph'][int(m.group(1))]=(float(m.group(3)),float(m.group(6)),float(m.group(7)),int(m.group(8)),int(m.group(9)))
for i,f in enumerate(frames[-16:]):
    tot=sum(v[0] for v in f['ph'].values())
    s=' '.join(f"{k}:{v[0]:.0f}/{v[2]:.0f}({v[4]})" for k,v in sorted(f['ph'].items()) if k<6)
    st=re.search(r'ring1=(\d+) need_ring2=(\d+) need_ring3=(\d+) fine_failed=(\d+) exhaustive=(\d+) own_empty=(\d+)', f['stats'])
    print(f"frame {i:2d}: sum of spans {tot:6.1f} us | it:span/maxB(misses) {s} | {st.group(0) if st else ''}")
PY
