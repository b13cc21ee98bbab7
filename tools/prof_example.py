import sys, os, torch, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/examples')
import train_light_synthetic as T
hist, dt = T.train(bn=16, steps=13, verbose=False)
print("ms/step after warm-up: %.2f" % (dt*1e3))
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CUDA, ProfilerActivity.CPU]) as prof:
    T.train(bn=16, steps=4, verbose=False)
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=18, max_name_column_width=60))
