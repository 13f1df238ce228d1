#!/bin/bash
# FETCH_SIZE calibration on gfx950 (run on the GPU box): known bytes of a streaming read and of random 32-byte probes against the counter
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/fetch_calib; rm -rf $out; mkdir -p $out
hipcc --offload-arch=gfx950 -O3 -o $out/fetch_calib tools/fetch_calib.hip || exit 1
$out/fetch_calib > $out/known.txt || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- $out/fetch_calib > /dev/null 2> $out/fetch.err
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/tcc -- $out/fetch_calib > /dev/null 2> $out/tcc.err
python3 - <<'PY'
import csv, glob, json
known = {}
for l in open('gpurun_out/fetch_calib/known.txt'):
    p = l.split()
    if p[0] == 'k_stream': known['k_stream'] = int(p[2])
    if p[0] == 'k_probe': known['k_probe'] = int(p[4]); known['probes'] = int(p[2])
res = {}
for f in glob.glob('gpurun_out/fetch_calib/*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        res.setdefault(k, {})[r['Counter_Name']] = res.get(k, {}).get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
out = {"known_bytes": known, "counters": res, "factors": {}}
for k in ('k_stream', 'k_probe'):
    kk = [x for x in res if k in x]
    if kk and 'FETCH_SIZE' in res[kk[0]]:
        rep = res[kk[0]]['FETCH_SIZE'] * 1024
        out["factors"][k] = {"fetch_size_bytes": rep, "known_bytes": known[k], "known_over_reported": known[k] / rep if rep else None}
json.dump(out, open('gpurun_out/fetch_calib/summary.json', 'w'), indent=1)
print(json.dumps(out["factors"], indent=1))
PY
