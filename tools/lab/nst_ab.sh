set -u
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out/nst
export TMPDIR=/tmp PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_interpn.py tests/test_gpu_dice_cce.py tests/test_gpu_deferred.py tests/test_gpu_vxm.py -q -p no:cacheprovider -x 2>&1 | tail -3
for r in 1 2 3; do
  echo "{\"lib\": \"nst3\"}" >> gpurun_out/nst/ab.jsonl; timeout 300 python tools/standalone_batch_probe.py 1 4 32 >> gpurun_out/nst/ab.jsonl 2>/dev/null
  echo "{\"lib\": \"nst2\"}" >> gpurun_out/nst/ab.jsonl; NEURITE_AMD_LIB=$GRAFT_REPO_ROOT/tools/lab/libnrt_fused_nst2.so timeout 300 python tools/standalone_batch_probe.py 1 4 32 >> gpurun_out/nst/ab.jsonl 2>/dev/null
done
cat gpurun_out/nst/ab.jsonl | cut -c1-330
