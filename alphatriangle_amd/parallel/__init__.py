"""Multi-GPU runtime: one process per GPU over torch.distributed (RCCL on ROCm, gloo for CPU tests)."""
