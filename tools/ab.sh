# A/B of an environment knob on the bench step: tools/ab.sh VAR "v1 v2 ..." [repeats]  (run on the GPU box)
VAR=$1; VALS=$2; REP=${3:-2}
for r in $(seq 1 $REP); do
  for v in $VALS; do
    ms=$(env $VAR=$v python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-fusion-probe 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$VAR=$v ms_per_step $ms"
  done
done
