/*
 * aas_warpctc.h - the warp-ctc C ABI, exported by libaas_hip.so with warp-ctc's own names and signatures.
 *
 * The reference calls `warpctc_pytorch.CTCLoss` (Speech_enhancement_by_AAS/trainer_AAS.py:10,62,168,349;
 * trainer_acoustic.py:10,56,132; AM_training/train.py:10,151,319): SeanNaren's PyTorch binding of baidu-research/warp-ctc,
 * whose native library exports the four functions below (warp-ctc include/ctc.h; no version is pinned by the reference,
 * README.md:8).  warp-ctc's sources are NOT in /root/reference: the declarations restate its published header so that the
 * binding's `gpu_ctc` (which calls get_workspace_size / compute_ctc_loss with `ctcOptions{loc = CTC_GPU, stream, blank_label}`)
 * links against libaas_hip.so unchanged.  They are thin wrappers over aas_ctc_get_workspace_size / aas_compute_ctc_loss
 * (aas_hip.h); the time extent of the activation tensor, which warp-ctc derives as max(input_lengths), is derived the same way.
 *
 * GPU location only: activations / gradients / workspace are DEVICE pointers, flat_labels / label_lengths / input_lengths /
 * costs HOST pointers, exactly as in warp-ctc's GPU path; `options.loc = CTC_CPU` returns CTC_STATUS_EXECUTION_FAILED (this
 * library has no CPU path).  compute_ctc_loss is synchronous on `options.stream`, as warp-ctc's is (costs are on the host).
 */
#ifndef AAS_WARPCTC_H
#define AAS_WARPCTC_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    CTC_STATUS_SUCCESS = 0,
    CTC_STATUS_MEMOPS_FAILED = 1,
    CTC_STATUS_INVALID_VALUE = 2,
    CTC_STATUS_EXECUTION_FAILED = 3,
    CTC_STATUS_UNKNOWN_ERROR = 4
} ctcStatus_t;

typedef enum { CTC_CPU = 0, CTC_GPU = 1 } ctcComputeLocation;

/* warp-ctc's `CUstream stream` member is a HIP stream here (hipStream_t as void*) */
struct ctcOptions {
    ctcComputeLocation loc;
    union {
        unsigned int num_threads;
        void* stream;
    };
    int blank_label;
};

int get_warpctc_version(void);
const char* ctcGetStatusString(ctcStatus_t status);

/* activations [maxT, minibatch, alphabet_size] pre-softmax (device); gradients same shape or NULL (device); costs[minibatch]
 * (host) = -log p(labels_n | activations[:input_lengths[n], n]); gradients are wrt the PRE-softmax activations, zero for
 * t >= input_lengths[n]. */
ctcStatus_t compute_ctc_loss(const float* const activations, float* gradients, const int* const flat_labels,
                             const int* const label_lengths, const int* const input_lengths, int alphabet_size, int minibatch,
                             float* costs, void* workspace, struct ctcOptions options);

ctcStatus_t get_workspace_size(const int* const label_lengths, const int* const input_lengths, int alphabet_size, int minibatch,
                               struct ctcOptions info, size_t* size_bytes);

#ifdef __cplusplus
}
#endif
#endif
