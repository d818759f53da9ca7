python bench.py --no-extras --no-cpu --option arena_probe=32 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
a=d['config']['arena_probe'][0]
print('probe32: frac %.4f  ms %.4f  setup_s %.3f  candidates %d  probe_us %.1f fill_us %.1f' % (d['roofline']['frac'], d['ms_per_step'], d['config']['arena_setup_s'], a['candidates'], a['probe_us'], a['fill_us']))"
python bench.py --no-extras --no-cpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
a=d['config']['arena_probe'][0]
print('probe8 : frac %.4f  ms %.4f  setup_s %.3f  candidates %d  probe_us %.1f fill_us %.1f' % (d['roofline']['frac'], d['ms_per_step'], d['config']['arena_setup_s'], a['candidates'], a['probe_us'], a['fill_us']))"
