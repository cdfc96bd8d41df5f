"""The bench's truncate block alone (for rocprofv3 --kernel-trace --stats): fused apply_compress and exact compress!(apply)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import qilaplace_jl_amd as qil
ctx = qil.default_context()
print(json.dumps(bench.truncate_block(qil, ctx, reps=2)))
