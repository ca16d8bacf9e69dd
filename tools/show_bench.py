import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: print(l[:300]); continue
    print('step', d.get("ms_per_step"), d.get('value'))
    for k in ('acoustic','semantic_m','semantic_s','acoustic_decode','files'):
        if k in d:
            v=d[k]
            print(k, v.get('ms_per_step'), v.get('value'), 'pinned', v.get('checksum_pinned'), {g:x['ms_per_step'] for g,x in v.get('breakdown',{}).items()})
            if k=='acoustic_decode': print('  roofline', v.get('roofline'), v.get('error'))
            if k=='files': print('  ', json.dumps(v)[:1500])
