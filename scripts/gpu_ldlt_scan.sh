# LDL^T micro-benchmark at the condensed IEEE-118 order over the schedule switches (one line per setting)
N=${1:-2069}; B=${2:-64}
for pad in 0 20000 49000; do
  echo -n "TRAIL_PAD=$pad  "; SQPHIP_TRAIL_PAD=$pad timeout -k 10 120 python scripts/gpu_ldlt_bench.py $N $B 10 || exit 1
  echo -n "TRAIL_PAD=$pad NO_LOOKAHEAD "; SQPHIP_NO_LOOKAHEAD=1 SQPHIP_TRAIL_PAD=$pad timeout -k 10 120 python scripts/gpu_ldlt_bench.py $N $B 10 || exit 1
done
for pad in 0 20000; do echo -n "N=2813 TRAIL_PAD=$pad  "; SQPHIP_TRAIL_PAD=$pad timeout -k 10 120 python scripts/gpu_ldlt_bench.py 2813 $B 10 || exit 1; done
