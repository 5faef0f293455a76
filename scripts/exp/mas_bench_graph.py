import sys, json; sys.path.insert(0, '/root/repo')
import torch, bench
print(json.dumps(bench.bench_mas(torch.device("cuda:0")), indent=1))
