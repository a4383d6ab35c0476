"""Which `s_waitcnt vmcnt` sit inside the store loop of each rollout / step kernel?  vmcnt retires in order, so ANY vector-memory wait
in a loop that stores observation rows also waits for the rows stored before it -- and the compiler places such waits where a loaded
value is first USED: a value loaded before the loop and first used in a rare branch of the loop (the episode counter in the reset
branch), or the join behind a rare path that loads, silently becomes a `vmcnt(0)` in the loop (round 3: both happened).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -fno-strict-aliasing --cuda-device-only -S -o /tmp/snac.s snac_amd/csrc/k_roll2d.hip   (any k_*.hip unit)
    python tools/isa_loop_waits.py [name fragments ...]
prints, per kernel: the loop with the most stores, its store / load instruction counts and the vmcnt waits inside it."""
import re, sys
from collections import Counter
txt = open('/tmp/snac.s').read()
names = re.findall(r'^(_ZN\S*?(?:k_rollout|k_step)\S*?):', txt, re.M)
seen = set()
for name in names:
    if name in seen: continue
    seen.add(name)
    if len(sys.argv) > 1 and not any(k in name for k in sys.argv[1:]): continue
    start = txt.index('\n' + name + ':')
    end = txt.index('s_endpgm', start)
    lines = txt[start:end].split('\n')
    def loops():
        cur = None
        for l in lines:
            m = re.search(r'Header=(BB\d+_\d+)', l)
            if m: cur = m.group(1)
            m2 = re.match(r'\.L(BB\d+_\d+):.*Loop Header', l)
            if m2: cur = m2.group(1)
            elif l.startswith('.LBB') and 'in Loop' not in l and 'Loop Header' not in l: cur = None
            yield cur, l
    c = Counter()
    for cur, l in loops():
        if 'global_store' in l and cur: c[cur] += 1
    if not c: continue
    main = c.most_common(1)[0][0]
    w = [l.strip().replace('s_waitcnt ', '') for cur, l in loops() if cur == main and 's_waitcnt vmcnt' in l]
    ld = sum(1 for cur, l in loops() if cur == main and 'global_load' in l)
    print(name[22:75], main, 'stores', c[main], 'loads', ld, w)
