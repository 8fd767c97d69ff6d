import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
t = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
def tb(s): return [x for x in t if s in x][0]
kd, ks, pe, pi = tb('kernel_dispatch'), tb('kernel_symbol'), tb('pmc_event'), tb('info_pmc')
cols = [r[1] for r in c.execute('pragma table_info(%s)' % pe)]
q = ("select s.kernel_name, p.name, sum(e.value), count(distinct d.id), sum(d.end-d.start)/1e3/count(distinct e.pmc_id) from %s e join %s p on e.pmc_id=p.id "
     "join %s d on e.event_id=d.event_id join %s s on d.kernel_id=s.id group by s.kernel_name, p.name" % (pe, pi, kd, ks))
res = {}
for name, cn, v, n, us in c.execute(q):
    res.setdefault(name, {})[cn] = (v, n)
for name, d in res.items():
    if 'conv' not in name: continue
    n = list(d.values())[0][1]
    print(name[:70], 'dispatches', n)
    wc = d.get('SQ_WAVE_CYCLES', (0, 1))[0]
    for cn, (v, _) in sorted(d.items()):
        print('   %-28s %14.0f  %6.1f%% of wave cycles' % (cn, v / n, 100.0 * v / wc if wc else 0))
