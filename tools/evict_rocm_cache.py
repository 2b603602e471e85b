"""Drop the page cache of the ROCm math libraries (posix_fadvise DONTNEED) so that the next process
starts as on a fresh box (cold library loads widen first-use races between host threads)."""
import glob, os
pats = ["/opt/rocm/lib/librocsolver.so*", "/opt/rocm/lib/librocblas.so*", "/opt/rocm/lib/libhipblaslt.so*",
        "/opt/rocm/lib/rocblas/library/*gfx950*", "/opt/rocm/lib/hipblaslt/library/*gfx950*",
        "/opt/rocm/lib/libamdhip64.so*", "/opt/rocm/lib/librocsparse.so*"]
n = b = 0
for p in pats:
    for f in glob.glob(p):
        try:
            fd = os.open(f, os.O_RDONLY)
            b += os.fstat(fd).st_size
            os.posix_fadvise(fd, 0, 0, os.POSIX_FADV_DONTNEED)
            os.close(fd)
            n += 1
        except OSError:
            pass
print(f"evicted {n} files, {b / 2**20:.0f} MiB")
