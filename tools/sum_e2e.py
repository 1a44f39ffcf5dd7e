import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); e=d["e2e"]
print(sys.argv[1], d["value"], e["gemm"]["odirect"]["seconds_all"], e["gemm"]["buffered"]["seconds_all"])
