run() { python bench.py --steps 10 --no-cpu "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(int(d['value']), round(d['roofline']['frac'],3))"; }
echo default; run
echo "mask2 lanes2 chunk256"; ABC_HIP_CU_MASK=2 run
echo "mask2 lanes2 chunk128"; ABC_HIP_CU_MASK=2 ABC_HIP_CHUNK=128 run
echo "mask2 lanes2 chunk512 b2048"; ABC_HIP_CU_MASK=2 ABC_HIP_CHUNK=512 run --batch 2048
echo "mask4 lanes4 chunk128"; ABC_HIP_CU_MASK=4 ABC_HIP_LANES=4 ABC_HIP_CHUNK=128 run
echo "mask2 lanes4 chunk128"; ABC_HIP_CU_MASK=2 ABC_HIP_LANES=4 ABC_HIP_CHUNK=128 run
echo default; run
