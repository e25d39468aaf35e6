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
