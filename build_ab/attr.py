import torch
p=torch.cuda.get_device_properties(0)
print(p)
print("shared_memory_per_block", getattr(p,"shared_memory_per_block",None), "optin", getattr(p,"shared_memory_per_block_optin",None), "per_mp", getattr(p,"shared_memory_per_multiprocessor",None))
