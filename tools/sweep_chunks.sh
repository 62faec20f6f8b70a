run() { python bench.py --steps 10 --no-cpu "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(int(d['value']), round(d['roofline']['frac'],3))"; }
echo "default b1024"; run
echo "b2048"; run --batch 2048
echo "b2048 chunk512"; ABC_HIP_CHUNK=512 run --batch 2048
echo "b4096 chunk512"; ABC_HIP_CHUNK=512 run --batch 4096
echo "b2048 chunk128 lanes4"; ABC_HIP_CHUNK=128 ABC_HIP_LANES=4 run --batch 2048
echo "b2048 chunk256 lanes3"; ABC_HIP_CHUNK=256 ABC_HIP_LANES=3 run --batch 2048
echo "b1024 chunk128 lanes2"; ABC_HIP_CHUNK=128 run
echo "b1024 chunk1024 lanes1"; ABC_HIP_CHUNK=1024 ABC_HIP_LANES=1 run
