python -m pytest tests/test_gpu_models.py -k "split_batch" -q 2>&1 | tail -3
for j in 1 0; do for w in x3d_m mvit_b_32x3 x3d_l slowfast_r50; do for k in 1 2 3 4; do
python bench.py --tune split_joint_graph=$j --workload $w --streams $k --no-cpu-baseline --no-secondary --no-sustained 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('joint=$j $w streams', d['config']['streams'], d['value'], d['ms_per_step'])"
done; done; done
