// warp-ctc's exported C ABI (include/aas_warpctc.h) over this library's CTC kernels: the native entry points behind
// warpctc_pytorch.CTCLoss (reference call sites: Speech_enhancement_by_AAS/trainer_AAS.py:168, AM_training/train.py:319).
#include "common.h"

#include "../../include/aas_warpctc.h"

extern "C" int get_warpctc_version(void) { return 2; }

extern "C" const char* ctcGetStatusString(ctcStatus_t status) {
    switch (status) {
        case CTC_STATUS_SUCCESS: return "no error";
        case CTC_STATUS_MEMOPS_FAILED: return "cuda memcpy or memset failed";
        case CTC_STATUS_INVALID_VALUE: return "invalid value";
        case CTC_STATUS_EXECUTION_FAILED: return "execution failed";
        default: return "unknown error";
    }
}

static int max_len(const int* v, int n) {
    int m = 0;
    for (int i = 0; i < n; ++i) m = v[i] > m ? v[i] : m;
    return m;
}

extern "C" ctcStatus_t get_workspace_size(const int* const label_lengths, const int* const input_lengths, int alphabet_size,
                                          int minibatch, struct ctcOptions info, size_t* size_bytes) {
    if (label_lengths == nullptr || input_lengths == nullptr || size_bytes == nullptr || alphabet_size <= 0 || minibatch <= 0)
        return CTC_STATUS_INVALID_VALUE;
    if (info.loc != CTC_GPU) return CTC_STATUS_EXECUTION_FAILED;     // no CPU path in this library
    const int max_T = max_len(input_lengths, minibatch);
    return aas_ctc_get_workspace_size(label_lengths, input_lengths, alphabet_size, minibatch, max_T > 0 ? max_T : 1, size_bytes) == 0
               ? CTC_STATUS_SUCCESS
               : CTC_STATUS_INVALID_VALUE;
}

extern "C" ctcStatus_t compute_ctc_loss(const float* const activations, float* gradients, const int* const flat_labels,
                                        const int* const label_lengths, const int* const input_lengths, int alphabet_size,
                                        int minibatch, float* costs, void* workspace, struct ctcOptions options) {
    if (activations == nullptr || flat_labels == nullptr || label_lengths == nullptr || input_lengths == nullptr ||
        costs == nullptr || workspace == nullptr || alphabet_size <= 0 || minibatch <= 0)
        return CTC_STATUS_INVALID_VALUE;
    if (options.loc != CTC_GPU) return CTC_STATUS_EXECUTION_FAILED;
    if (options.blank_label < 0 || options.blank_label >= alphabet_size) return CTC_STATUS_INVALID_VALUE;
    const int max_T = max_len(input_lengths, minibatch);
    const int rc = aas_compute_ctc_loss(options.stream, activations, gradients, flat_labels, label_lengths, input_lengths, alphabet_size,
                                        minibatch, max_T > 0 ? max_T : 1, costs, workspace, options.blank_label);
    return rc == 0 ? CTC_STATUS_SUCCESS : (rc == 1 ? CTC_STATUS_INVALID_VALUE : CTC_STATUS_EXECUTION_FAILED);
}
