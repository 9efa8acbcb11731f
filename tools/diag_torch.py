import sys, os, ctypes
mode = sys.argv[1]
sys.path.insert(0, '/root/repo')
if mode == 'torch_first':
    import torch
    print('torch avail', torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.zeros(4, device='cuda'); print('alloc ok')
    from camkifu_amd import capi
    c = capi.Context(0); print('ctx ok after torch')
elif mode == 'lib_first':
    from camkifu_amd import capi
    c = capi.Context(0); print('ctx ok')
    import torch
    print('torch avail', torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.zeros(4, device='cuda'); print('alloc ok')
elif mode == 'import_then_lib_then_cuda':
    import torch
    from camkifu_amd import capi
    c = capi.Context(0); print('ctx ok')
    print('torch avail', torch.cuda.is_available(), torch.cuda.device_count())
    x = torch.zeros(4, device='cuda'); print('alloc ok')
with open('/proc/self/maps') as f:
    libs = sorted(set(l.split()[-1] for l in f if 'amdhip' in l or 'hsa-runtime' in l))
print(libs)
