# tools/_ab.sh — steady-state A/B of this tree against an OLDER tree of this repo in one gpurun call (profiles/r06_steady_ab_vs_r05.log).
# The old tree (its own tools, its own built nrays_amd/lib/libnrays_hip.so — an older ABI cannot be loaded by this tree's Python) is expected under ./_r05/ :
#   git archive <commit> | tar -x -C /tmp/old && (cd /tmp/old && python -c "import __graft_entry__ as g; g.build()") && cp -r /tmp/old/{nrays_amd,tools,oracle,scenes,include,__graft_entry__.py} _r05/
# (_r05/ is listed in .git/info/exclude; delete it afterwards: every gpurun push carries it.)
for i in 1 2; do
 echo "== r05"; (cd _r05 && python tools/kbench.py --scenes sponza,balls,hairball,primitives,sponza8 --steps 100 2>&1 | grep scene | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['scene'], d['ms'], d['primary_ms'])")
 echo "== r06"; python tools/kbench.py --scenes sponza,balls,hairball,primitives,sponza8 --steps 100 2>&1 | grep scene | python -c "
import sys, json
for l in sys.stdin:
    d=json.loads(l); print(d['scene'], d['ms'], d['primary_ms'])"
done
