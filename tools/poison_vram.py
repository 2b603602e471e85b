"""Fill (most of) the GPU's memory with a byte pattern and exit: the next process then finds that
pattern in every buffer it allocates and has not written (VRAM is not cleared between processes
on this stack), so a read of uninitialised device memory shows as a deterministic failure instead
of depending on what the previous process left behind.
usage: poison_vram.py [byte=255] [GiB=240]"""
import sys, torch
byte = int(sys.argv[1]) if len(sys.argv) > 1 else 255
gib = int(sys.argv[2]) if len(sys.argv) > 2 else 240
keep = []
try:
    for _ in range(gib // 4):
        t = torch.empty(4 << 30, dtype=torch.uint8, device="cuda:0")
        t.fill_(byte)
        keep.append(t)
except RuntimeError:
    pass
torch.cuda.synchronize()
print(f"poisoned {4 * len(keep)} GiB with byte {byte}")
