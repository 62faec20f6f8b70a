# throughput of the headline bench under different chunk / lane plans (ABC_HIP_CHUNK, ABC_HIP_LANES)
run() { python bench.py --steps 10 --no-cpu "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(int(d['value']), round(d['roofline']['frac'],3))"; }
echo "default"; run
for cfg in "64 2" "32 2" "32 4" "16 4" "64 4" "128 2" "512 2"; do set -- $cfg; echo "chunk $1 lanes $2"; ABC_HIP_CHUNK=$1 ABC_HIP_LANES=$2 run; done
echo "default"; run
