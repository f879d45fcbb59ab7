cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --force-dist --exchange flat --steps 66 --warmup 2 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['scaling'], d['config']['launch'][:60], d['config']['exchange'], d['comm'])"
timeout 600 python bench.py --force-dist --steps 66 --warmup 2 --no-extras --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['scaling'], d['config']['launch'][:60], d['config']['exchange'], d['comm'])"
