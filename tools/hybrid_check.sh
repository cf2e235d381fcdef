#!/bin/bash
# tools/hybrid_check.sh -- `oswald -m 0` against `-m 1` (hybrid) on a C2-size database: same report, wall times (GPU box)
python tools/cli_e2e.py 100000 /tmp/osw_e2e > /dev/null 2>&1 || true
for m in 0 1; do
  t0=$(date +%s.%N)
  oswald_amd/oswald -O search -m $m -c 16 -q /tmp/osw_e2e/q.fasta -d /tmp/osw_e2e/db > /tmp/out_$m.txt
  t1=$(date +%s.%N)
  python3 -c "print(\"mode $m wall\", round($t1 - $t0, 3), \"s\")"
  grep "Search time\|Search speed\|estimated\|Test DB" /tmp/out_$m.txt
done
diff <(grep -v "Search\|estimated\|Test DB\|CPU threads" /tmp/out_0.txt) <(grep -v "Search\|estimated\|Test DB\|CPU threads" /tmp/out_1.txt) && echo "reports identical"
