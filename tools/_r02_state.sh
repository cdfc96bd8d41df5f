#!/bin/bash
R=$GRAFT_REPO_ROOT
python -m pytest $R/tests -x -q -m gpu 2>&1 | tail -3
python $R/tools/_truncate_block.py 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k:(round(v,1) if isinstance(v,float) and v>1 else v) for k,v in d.items() if k.endswith('ms') or 'err' in k or 'bonds' in k})"
python $R/tools/_compress_time.py 2>&1 | grep compress
python $R/tools/_prof_encode30.py 2>&1 | tail -1
python $R/tools/_fuzz_product_compress.py 60 2>&1 | tail -1
