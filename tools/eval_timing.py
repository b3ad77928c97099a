import sys, json, torch
sys.path.insert(0,'/root/repo')
import bench
print(json.dumps(bench.eval_timing(torch.device('cuda:0')), indent=1))
