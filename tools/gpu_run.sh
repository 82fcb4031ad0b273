export TMPDIR=/tmp; mkdir -p gpurun_out
(timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "extract or orient or timed or pipe or findpoints" > gpurun_out/r02_pytest_m.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r02_pytest_m.log); tail -3 gpurun_out/r02_pytest_m.log
python - <<'PY'
import json
d=json.load(open('gpurun_out/parity_report.json'))
mo=0; md=0; mc=1; ms=0; no=0
for k,v in sorted(d.items()):
    if 'desc_outliers' in v:
        mo=max(mo,v.get('orient_maxdiff_deg_inliers',0)); md=max(md,v.get('desc_maxabs_same_orient',0)); mc=min(mc,v.get('desc_min_cos_same_orient',1)); ms=max(ms,v.get('scale_relerr_max',0)); no+=v['desc_outliers']+v['orient_outliers']
print("ALL: orient max deg", mo, "desc maxabs", md, "min cos", mc, "scale relerr", ms, "outliers", no)
PY
