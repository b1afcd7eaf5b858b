"""One line per bench JSON: value, ms/step, sweeps, roofline fraction, factorisations per QP."""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, round(d["value"], 1), round(d["ms_per_step"], 2), d["config"]["sweeps"], round(d["roofline"]["frac"], 4),
          round(d["config"]["factorisations_per_qp"], 2))
