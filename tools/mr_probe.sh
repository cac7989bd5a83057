cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_more.py -m gpu -q -x -k "two_ranks_match or four_and_eight or eam_and_sp_on or two_ghost_layers_several or exchange_all or loopback" 2>&1 | tail -15
for n in 2 8; do
python bench.py --gpus $n --steps 40 --warmup 20 --size 20 --equil 20 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][0])
print('N', d['n_gpus'], 'value', d['value'], 'syncs/rebuild', d['host_syncs_per_rebuild'], 'syncs/step', d['host_syncs_per_step'], 'transport syncs/step', d['host_transport_syncs_per_step'], 'halo B/step', d['halo_bytes_per_step'], d['phases_s_max'])"
done
