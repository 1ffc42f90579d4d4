OUT=$GRAFT_REPO_ROOT/gpurun_out/r6v21; mkdir -p $OUT
for o in 0 1 0 1; do BEVOPS_TSGEMM_ORDER=$o python tools/tsgemm_time.py 2>/dev/null | sed "s/^{/{\"order\": $o, /"; done > $OUT/tsgemm_order_ab.jsonl
python - <<'PY'
import json,collections
d=collections.defaultdict(lambda: {0:[],1:[]})
for l in open('gpurun_out/r6v21/tsgemm_order_ab.jsonl'):
    r=json.loads(l); d[r['layer']][r['order']].append(r['us_tsgemm']); d[r['layer']]['lib']=r['us_lib']
for k,v in d.items(): print(k, 'lib', v['lib'], 'order0', v[0], 'order1', v[1])
PY
