// Exact brute-force cosine top-k over the CLIP-embedding database held in HBM (host interface).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct KnnDb {
    long long n = 0; int dim = 0;
    void* dbn = nullptr;        // fp16 [n_pad, dim] normalised rows (n_pad = n rounded up to 256; tail rows zero)
    void* raw = nullptr;        // raw embeddings in the caller's dtype [n, dim]
    int raw_dtype = 0;          // 0 fp16, 1 fp32
    // search scratch
    void* scratch = nullptr; size_t scratch_bytes = 0;
    void* zero_page = nullptr;  // 256 zero bytes on the database's device (source of padding lanes)
    int last_fallbacks = 0;     // 1 if the last search needed the exact fallback pass for at least one query
};

// all return nullptr on success, or a static error string
const char* knn_load(KnnDb& db, const void* emb, long long n, int dim, int dtype, int is_device, hipStream_t st);
const char* knn_search(KnnDb& db, const float* q, int b, int k, uint32_t* idx_out, float* score_out, double* score64_out, hipStream_t st);
const char* knn_gather(KnnDb& db, const uint32_t* idx, long long n_idx, float* out, hipStream_t st);
void knn_free(KnnDb& db);
