#!/bin/bash
# per-kernel in-situ averages of the one-clip workload under several environment settings on one box:
#   bash tools/trace_env.sh OUT "Q3A_X=0" "Q3A_X=7" ...      (each argument: one env assignment list, or the word base)
out=$1; shift
mkdir -p "$out"
export TMPDIR=/tmp
for setting in "$@"; do
  tag=$(echo "$setting" | tr ' =' '__')
  d=$(mktemp -d /tmp/q3a_trace_XXXX)
  if [ "$setting" = base ]; then envs=""; else envs="$setting"; fi
  (cd /tmp && env $envs rocprofv3 --kernel-trace --stats -d "$d" -o t -- python "$OLDPWD/bench.py" --inner ${TRACE_ARGS:---preset 0.6b --batch 1 --seconds 30 --new-tokens 100 --steps 3 --warmup 1} > /dev/null 2>&1)
  python - "$d" "$setting" <<'PY' > "$out/trace_$tag.txt"
import glob, sqlite3, sys, re
db = glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True)[0]
rows = sqlite3.connect(db).execute("select name, start, end from kernels order by start").fetchall()
per = {}
for n, s, e in rows:
    n = re.sub(r"q3a::\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n).split("(")[0]
    per.setdefault(n, []).append((e - s) / 1e3)
print("#", sys.argv[2])
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    kept = v[len(v) // 4:]  # drop the warm-up pass
    print(f"{k[:70]:70s} {len(kept):6d} {sum(kept)/len(kept):8.2f} us")
PY
  head -8 "$out/trace_$tag.txt"
  rm -rf "$d"
done
