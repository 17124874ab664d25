#!/bin/bash
# tools/derep_run.sh — d = 0 at scale on the GPU box: 5 M raw reads over 500 k distinct
# sequences, this build vs oracle/_ref/swarm (if present), outputs compared byte for byte.
TIMEFORMAT="%R s"
gcc -O2 -o tools/gen_amplicons tools/gen_amplicons.c -lm
tools/gen_amplicons 500000 150 5 1 0 /tmp/u.fa
python tools/make_reads.py /tmp/u.fa ${1:-5000000} 1 /tmp/reads.fa
echo -n "ours (1st, includes GPU context creation): "; time swarm_amd/bin/swarm -d 0 -o /tmp/o1 -w /tmp/w1 -s /tmp/s1 -l /tmp/l1 /tmp/reads.fa
echo -n "ours (2nd): "; time swarm_amd/bin/swarm -d 0 -o /tmp/o1 -w /tmp/w1 -s /tmp/s1 -l /tmp/l1 /tmp/reads.fa
tail -3 /tmp/l1
if [ -x oracle/_ref/swarm ]; then
  echo -n "reference: "; time oracle/_ref/swarm -d 0 -o /tmp/o2 -w /tmp/w2 -s /tmp/s2 -l /tmp/l2 /tmp/reads.fa
  cmp /tmp/o1 /tmp/o2 && cmp /tmp/w1 /tmp/w2 && cmp /tmp/s1 /tmp/s2 && echo IDENTICAL
fi
