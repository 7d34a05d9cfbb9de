import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        h=d.get('host_step_intervals') or {}
        ip=d.get('iota_proof') or {}
        print(f.split('/')[-1], 'ms %.3f tagged %s | min %.3f med %.3f max %.3f 2nd %.3f first %.3f gc %s' % (d['ms_per_step'], ip.get('ms_per_step_with_producer_tagged_offsets'), h.get('min_ms',0), h.get('median_ms',0), h.get('max_ms',0), h.get('second_max_ms',0), h.get('first_ms',0), h.get('python_gc_collections_in_timed_region')))
    except Exception as e: print(f, 'ERR', e)
