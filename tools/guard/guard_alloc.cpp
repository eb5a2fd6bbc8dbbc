// Guard allocator for tests (called through ctypes by tools/guard_soak.py): every allocation gets its own HIP virtual-memory
// mapping with an UNMAPPED granule on both sides, and the buffer is placed so that its END (DC_GUARD_MODE=end, default) or its
// START (DC_GUARD_MODE=start) coincides with the edge of the mapping (end mode: up to the 512-byte alignment torch's kernels need): a
// kernel that reads or writes past the buffer faults deterministically ("Memory access fault by GPU") instead of landing in a neighbouring cached block.
// Test infrastructure only (tests/test_gpu_guard.py, tools/guard_soak.py); not part of the product library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>

namespace {
struct Rec { void* va; size_t reserve; size_t mapped; hipMemGenericAllocationHandle_t h; };
std::mutex g_mu;
std::unordered_map<void*, Rec> g_recs;
size_t g_gran = 0;
int g_mode_start = -1, g_fill = -1;
size_t g_live = 0, g_peak = 0, g_count = 0;

void die(const char* what, hipError_t e) {
    std::fprintf(stderr, "dc_guard: %s failed: %s\n", what, hipGetErrorString(e));
    std::abort();
}
}  // namespace

extern "C" void* dc_guard_malloc(ssize_t size, int device, hipStream_t) {
    std::lock_guard<std::mutex> lk(g_mu);
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    if (!g_gran) {
        hipError_t e = hipMemGetAllocationGranularity(&g_gran, &prop, hipMemAllocationGranularityMinimum);
        if (e != hipSuccess) die("hipMemGetAllocationGranularity", e);
        const char* m = std::getenv("DC_GUARD_MODE");
        g_mode_start = (m && std::strcmp(m, "start") == 0) ? 1 : 0;
        const char* f = std::getenv("DC_GUARD_FILL");
        g_fill = f ? std::atoi(f) : 0;
        std::fprintf(stderr, "dc_guard: granularity %zu, mode %s, fill %d\n", g_gran, g_mode_start ? "start" : "end", g_fill);
    }
    // Alignment of the base: 16 bytes (DC_GUARD_ALIGN overrides), so an overrun is caught once it passes the buffer's end rounded up
    // to 16.  Only buffers this library's kernels see are allocated here (dotaclient_amd.engine.DEVICE_ALLOC_HOOK); torch's own
    // temporaries stay on torch's allocator, whose 512-byte alignment torch's kernels rely on.
    static const size_t align = [] { const char* a = std::getenv("DC_GUARD_ALIGN"); size_t v = a ? (size_t)std::atoll(a) : 16; return v < 16 ? 16 : v; }();
    size_t need = size <= 0 ? align : ((size_t)size + align - 1) / align * align;
    size_t mapped = (need + g_gran - 1) / g_gran * g_gran;
    Rec r;
    r.mapped = mapped;
    r.reserve = mapped + 2 * g_gran;
    hipError_t e = hipMemAddressReserve(&r.va, r.reserve, g_gran, nullptr, 0);
    if (e != hipSuccess) die("hipMemAddressReserve", e);
    e = hipMemCreate(&r.h, mapped, &prop, 0);
    if (e != hipSuccess) die("hipMemCreate", e);
    char* base = (char*)r.va + g_gran;
    e = hipMemMap(base, mapped, 0, r.h, 0);
    if (e != hipSuccess) die("hipMemMap", e);
    hipMemAccessDesc acc;
    std::memset(&acc, 0, sizeof(acc));
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    e = hipMemSetAccess(base, mapped, &acc, 1);
    if (e != hipSuccess) die("hipMemSetAccess", e);
    void* p = g_mode_start ? (void*)base : (void*)(base + mapped - need);
    if (g_fill) {                                   // 0xFF bytes: NaN as f32, 255 as u8, -1 as an index
        // DC_GUARD_FILL=1: the whole mapping (reads of uninitialised memory show up); =2: only the PADDING around the buffer (the up to
        // 4 KB before it in end mode / after it in start mode), the buffer itself zeroed - what then still fails reads outside its buffer
        e = hipMemset(base, 0xFF, mapped);
        if (e != hipSuccess) die("hipMemset", e);
        if (g_fill == 2) {
            e = hipMemset(p, 0, (size_t)(size <= 0 ? 0 : size));
            if (e != hipSuccess) die("hipMemset", e);
        }
    }
    g_recs[p] = r;
    g_live += mapped; g_count++;
    if (g_live > g_peak) g_peak = g_live;
    return p;
}

extern "C" void dc_guard_free(void* ptr, ssize_t, int, hipStream_t) {
    if (!ptr) return;
    (void)hipDeviceSynchronize();                   // no stream-ordered reuse here: nothing may still be running on the block
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_recs.find(ptr);
    if (it == g_recs.end()) { std::fprintf(stderr, "dc_guard: free of unknown pointer %p\n", ptr); return; }
    Rec r = it->second;
    g_recs.erase(it);
    char* base = (char*)r.va + g_gran;
    (void)hipMemUnmap(base, r.mapped);
    (void)hipMemRelease(r.h);
    (void)hipMemAddressFree(r.va, r.reserve);
    g_live -= r.mapped;
}

extern "C" void dc_guard_stats(size_t* live, size_t* peak, size_t* count) { *live = g_live; *peak = g_peak; *count = g_count; }
