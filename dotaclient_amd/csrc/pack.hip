// Host-side ingest helper: fills the page-locked staging buffers of a batch from the wire-format arrays of its
// rollouts (replaces the per-key slicing of /root/reference/optimizer.py:353-365).
//
// A batch is ~1 100 small 2-D copies (17 arrays per rollout, each [T, w] contiguous, into the column block
// [r0:r0+T, c:c+w] of a [rows, 483] / [rows, 65] / [rows, 10] staging buffer).  From Python that is one numpy call
// per copy under the GIL (~5 us each: 5.8 ms per 64 x 256 batch); here the whole list is one call, split over a few
// threads by bytes.  No device work: the H2D copies of the filled buffers stay with the caller (four per batch).
#include <string.h>
#include <thread>
#include <vector>
#include "kernels.h"

namespace dc {
namespace {

struct PackItem { const char* src; char* dst; int64_t rows, row_bytes, dst_stride; };

void run_items(const PackItem* it, int64_t first, int64_t last) {
    for (int64_t i = first; i < last; ++i) {
        const PackItem& p = it[i];
        if (p.src == nullptr) {                                  // zero pad (the staging buffers are reused)
            if (p.row_bytes == p.dst_stride) memset(p.dst, 0, (size_t)(p.rows * p.row_bytes));
            else for (int64_t r = 0; r < p.rows; ++r) memset(p.dst + r * p.dst_stride, 0, (size_t)p.row_bytes);
        } else if (p.row_bytes == p.dst_stride) {
            memcpy(p.dst, p.src, (size_t)(p.rows * p.row_bytes));
        } else {
            for (int64_t r = 0; r < p.rows; ++r) memcpy(p.dst + r * p.dst_stride, p.src + r * p.row_bytes, (size_t)p.row_bytes);
        }
    }
}

}  // namespace
}  // namespace dc

extern "C" int dc_pack_rows(const int64_t* src, const int64_t* dst, const int64_t* rows, const int64_t* row_bytes,
                            const int64_t* dst_stride, int64_t n_items, int n_threads) {
    using namespace dc;
    if (n_items < 0 || (n_items > 0 && (!src || !dst || !rows || !row_bytes || !dst_stride))) {
        set_error("dc_pack_rows: bad arguments", 1020);
        return 1020;
    }
    std::vector<PackItem> items((size_t)n_items);
    int64_t total = 0;
    for (int64_t i = 0; i < n_items; ++i) {
        if (rows[i] < 0 || row_bytes[i] < 0 || dst_stride[i] < row_bytes[i] || dst[i] == 0) {
            set_error("dc_pack_rows: bad item", 1021);
            return 1021;
        }
        items[(size_t)i] = PackItem{reinterpret_cast<const char*>(src[i]), reinterpret_cast<char*>(dst[i]), rows[i], row_bytes[i], dst_stride[i]};
        total += rows[i] * row_bytes[i];
    }
    int nt = n_threads < 1 ? 1 : (n_threads > 32 ? 32 : n_threads);
    if (total < (4 << 20)) nt = 1;                               // small batches: thread start costs more than the copy
    if (nt == 1) { run_items(items.data(), 0, n_items); return 0; }
    // contiguous item ranges of ~equal bytes (items of one rollout stay together: neighbouring destination rows)
    std::vector<int64_t> cut(1, 0);
    int64_t acc = 0;
    for (int64_t i = 0; i < n_items; ++i) {
        acc += rows[i] * row_bytes[i];
        if (acc * nt >= total * (int64_t)cut.size() && (int)cut.size() < nt) cut.push_back(i + 1);
    }
    cut.push_back(n_items);
    std::vector<std::thread> th;
    for (size_t k = 0; k + 2 < cut.size(); ++k) th.emplace_back(run_items, items.data(), cut[k], cut[k + 1]);
    run_items(items.data(), cut[cut.size() - 2], cut[cut.size() - 1]);
    for (auto& t : th) t.join();
    return 0;
}
