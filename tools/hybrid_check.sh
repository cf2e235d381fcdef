#!/bin/bash
# tools/hybrid_check.sh [NSEQ] -- `oswald -m 1` (hybrid, the CLI's default mode) against `-m 0` on a synthetic database: the
# report's speed lines at several -c, the phases of the hybrid run, and that the reports are the same (GPU box).
N=${1:-1000000}
T=/tmp/osw_e2e_$N
[ -f $T/db.info ] || python tools/cli_e2e.py $N $T > /dev/null 2>&1   # (tools/final_validation.sh leaves the 1 M database there)
oswald_amd/oswald -O search -m 0 -q $T/q.fasta -d $T/db > /tmp/hc_m0.txt
echo "-m 0: $(grep 'Search speed' /tmp/hc_m0.txt | tr -s '\t' ' ')"
for c in ${CS:-4 16 64}; do
  OSWALD_DEBUG_PHASES=1 oswald_amd/oswald -O search -m 1 -c $c -q $T/q.fasta -d $T/db > /tmp/hc_m1.txt 2> /tmp/hc_m1.err
  echo "-m 1 -c $c: $(grep 'estimated\|Search speed' /tmp/hc_m1.txt | tr -s '\t' ' ' | tr '\n' ';')"
  grep -E "hybrid:|host done|test thread|held|runs the host part" /tmp/hc_m1.err | sed 's/^/      /'
  diff <(grep -v "Search\|estimated\|Test DB\|CPU threads" /tmp/hc_m0.txt) <(grep -v "Search\|estimated\|Test DB\|CPU threads" /tmp/hc_m1.txt) > /dev/null && echo "      report identical to -m 0's"
done
