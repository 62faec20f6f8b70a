# experiments: lanes pinned to disjoint CU sets (ABC_HIP_CU_MASK) and started out of phase (ABC_HIP_LANE_OFFSET_US)
run() { python bench.py --steps 10 --no-cpu "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(int(d['value']), round(d['roofline']['frac'],3))"; }
echo default; run
for off in 150 300 450 600; do
echo "offset $off"; ABC_HIP_LANE_OFFSET_US=$off run
echo "mask2 offset $off"; ABC_HIP_CU_MASK=2 ABC_HIP_LANE_OFFSET_US=$off run
done
echo "mask2 offset 600 chunk512 b2048"; ABC_HIP_CU_MASK=2 ABC_HIP_LANE_OFFSET_US=600 ABC_HIP_CHUNK=512 run --batch 2048
echo "offset 600 chunk512 b2048"; ABC_HIP_LANE_OFFSET_US=600 ABC_HIP_CHUNK=512 run --batch 2048
echo "chunk512 b2048"; ABC_HIP_CHUNK=512 run --batch 2048
echo default; run
