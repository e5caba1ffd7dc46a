"""Internal package of the MI355X-native conditional-U-Net hot path.

``_lib``      ctypes binding of libwu_kernels.so (the C ABI of include/wu_kernels.h)
``layout``    NHWC tensor helpers (logical NCHW shape, channels-last memory, explicit pixel stride)
``functional`` torch.autograd.Function wrappers around the HIP kernels
``ddp``       one-process-per-GPU data parallelism: bucketed RCCL gradient all-reduce
"""
