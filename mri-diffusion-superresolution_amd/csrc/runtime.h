// Host-side runtime pieces shared by model.hip / capi.hip: device buffers, the bump arena, packed weights.
#pragma once
#include <map>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mrisr.h"
#include "common.h"

namespace mrisr {

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    int reserve(size_t n, bool zero) {
        if (n <= bytes) return 0;
        release();
        MRISR_CHECK_HIP(hipMalloc(&p, n));
        bytes = n;
        if (zero) MRISR_CHECK_HIP(hipMemset(p, 0, n));
        return 0;
    }
};

// Bump allocator with stack discipline.  In `dry` mode it only tracks the peak, so one dry pass of the
// forward sizes the workspace exactly and every later pass reproduces the same addresses (hipGraph-safe).
struct Arena {
    DevBuf buf;
    size_t off = 0, peak = 0;
    bool dry = false;
    void reset() { off = 0; }
    void* alloc(size_t bytes) {
        const size_t a = (off + 255) & ~(size_t)255;
        off = a + bytes;
        if (off > peak) peak = off;
        if (dry) return reinterpret_cast<void*>(0x1000 + a);  // non-null placeholder, never dereferenced
        if (off > buf.bytes) return nullptr;
        return static_cast<char*>(buf.p) + a;
    }
    size_t mark() const { return off; }
    void release(size_t m) { off = m; }
};

struct RawParam {
    std::shared_ptr<DevBuf> data;  // f32
    std::vector<int64_t> shape;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};

struct NormW {
    const float* g = nullptr;
    const float* b = nullptr;
    int c = 0;
    std::string name;  // state-dict module name (full-parameter training writes its affine gradients by name)
};
struct ConvW {  // packed [cout][ks*ks*cin] in compute dtype, bias f32
    void* w = nullptr;
    const float* b = nullptr;
    int cin = 0, cout = 0, ks = 3;
    std::string name;    // state-dict module name (the dgrad packer re-reads the raw f32 weight)
    void* wd = nullptr;  // training: dgrad filter bank [cin][ky'][kx'][cout] (taps flipped; 1x1: W^T [cin][cout]), compute dtype
    long long offW = -1, offB = -1;  // training (T2I-Adapter): offsets of weight / bias in the flat trainable vector
};
struct LinW {  // packed [n][k] in compute dtype, bias f32 (GEGLU: interleaved)
    void* w = nullptr;
    const float* b = nullptr;
    int n = 0, k = 0;
    // fused LoRA (peft): per fused module one rank-r adapter.  loraA [R = nmod*r][k] compute dtype, loraB f32 [n][r]
    // already scaled by lora_alpha/r; output column n uses z columns (n / secN) * r .. +r
    int r = 0, R = 0, secN = 1;
    void* loraA = nullptr;
    const float* loraB = nullptr;
    // fp8 copies for the row-panel kernel (cfg.fp8_linears; K = 320 / 640 only): e4m3 rows + one f32 scale per row
    void* w8 = nullptr;
    float* w_scale = nullptr;
    void* loraA8 = nullptr;       // [16][k] (rows >= R zero)
    float* loraA_scale = nullptr; // [16]
    // training
    std::vector<std::string> mod_names;  // the fused modules, in column order (e.g. attn1.to_q, to_k, to_v)
    std::vector<int> mod_lora;           // 1 if that module carries an adapter
    std::vector<long long> offA, offB;   // offsets of lora_A / lora_B of each module in the flat trainable vector (-1: none)
    void* wT = nullptr;                  // [k][n] compute dtype (dgrad operand)
    void* loraBT = nullptr;              // [R][n] compute dtype: (alpha/r) * B^T, zero outside the module's columns
    float* loraAT = nullptr;             // [k][R] f32: A^T (epilogue operand of the dgrad)
    float* loraB_rw = nullptr;           // writable alias of loraB (refreshed from the trainable vector)
};

struct Act {  // NHWC activation (or token rows when H*W is the token count)
    void* p = nullptr;
    int B = 0, H = 0, W = 0, C = 0;
    size_t rows() const { return (size_t)B * H * W; }
    size_t numel() const { return rows() * C; }
};

inline int dtype_size(int dt) { return dt == MRISR_F32 ? 4 : (dt == MRISR_I64 ? 8 : 2); }

}  // namespace mrisr
