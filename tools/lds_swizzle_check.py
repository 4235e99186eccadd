import itertools
groups = [list(range(0,4))+list(range(12,16))+list(range(20,28)),
          list(range(4,12))+list(range(16,20))+list(range(28,32)),
          list(range(32,36))+list(range(44,48))+list(range(52,60)),
          list(range(36,44))+list(range(48,52))+list(range(60,64))]
def conflicts(f, nbits=5):
    worst = 0; total = 0
    for qb in range(1<<nbits if nbits>4 else 32):
        for ks in range(2):
            for g in groups:
                slots = {}
                for lane in g:
                    l15 = lane & 15; g4 = lane >> 4
                    q = qb + l15
                    chunk = (4*ks+g4) ^ f(q)
                    addr = q*128 + chunk*16
                    slot = (addr >> 4) & 15
                    slots.setdefault(slot, set()).add(addr)
                w = max(len(v) for v in slots.values())
                total += sum(len(v)-1 for v in slots.values())
                worst = max(worst, w)
    return worst, total
# candidates: linear maps from low 5 bits of q to 3 bits
best = []
for rows in itertools.product(range(32), repeat=3):
    def f(q, rows=rows):
        v = 0
        for i, r in enumerate(rows):
            v |= (bin(q & r).count('1') & 1) << i
        return v
    w, t = conflicts(f)
    if t == 0:
        best.append(rows)
print(len(best), best[:20])
# specific ones
print('hx&7', conflicts(lambda q: q & 7))
print('(q>>1)&7', conflicts(lambda q: (q >> 1) & 7))
