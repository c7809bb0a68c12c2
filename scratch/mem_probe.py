"""HBM capacity as the runtime reports it (total / free bytes on a fresh box)."""
import torch
free, total = torch.cuda.mem_get_info(0)
print("total %d B = %.2f GiB = %.2f GB; free %d B = %.2f GiB" % (total, total / 2**30, total / 1e9, free, free / 2**30))
p = torch.cuda.get_device_properties(0)
print(p.name, p.total_memory, p.multi_processor_count)
