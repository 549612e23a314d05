"""dev: does this kernel track soft-dirty pages (CONFIG_MEM_SOFT_DIRTY: /proc/self/clear_refs '4' + pagemap bit 55)?  What an O(pages)
validation of unpinned host columns would stand on."""
import ctypes, mmap, os, struct
PAGE = 4096
m = mmap.mmap(-1, PAGE * 4)
m[0] = 1; m[PAGE] = 1; m[2 * PAGE] = 1
addr = ctypes.addressof(ctypes.c_char.from_buffer(m))
def bits(a, n):
    with open("/proc/self/pagemap", "rb") as f:
        f.seek((a // PAGE) * 8)
        d = f.read(8 * n)
    return [(struct.unpack_from("<Q", d, i * 8)[0] >> 55) & 1 for i in range(n)]
print(os.uname().release, "written, before clear:", bits(addr, 4))
try:
    open("/proc/self/clear_refs", "w").write("4")
    print("after clear:", bits(addr, 4))
    m[PAGE] = 2
    print("after writing page 1:", bits(addr, 4), "-> soft-dirty tracking", "WORKS" if bits(addr, 4)[1] == 1 else "NOT AVAILABLE")
except Exception as e:
    print("clear_refs:", e)
