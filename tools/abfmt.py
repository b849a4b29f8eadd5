import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('{'):
        d=json.loads(l); print(d['variant'], d['mode'], 'ms', d['ms'], 'fast', d['fast_ms'], 'chain', d['binning_chain_ms'], {k:d['stages'][k] for k in ('scan','duplicate','sort','tile_ranges')}, 'fast-stages', {k:(d['stages_fast'] or {}).get(k) for k in ('scan','duplicate','sort','tile_ranges')})
    elif not l.startswith('[gpurun] sending'): print(l)
