// flash_runtime.cpp -- level 3 (file-resident matrices). PLACEHOLDER for the first
// kernel bring-up; replaced by the real runtime.
#include "bof_hip.h"
#include "bof_internal.h"
extern "C" {
int bof_flash_gemm(char, char, char, uint64_t, uint64_t, uint64_t, float, float, bof_fptr, bof_fptr, bof_fptr, uint64_t, uint64_t, uint64_t, const bof_options *) { bof::set_error("not built yet"); return BOF_EINVAL; }
int bof_flash_csrmm(char, uint64_t, uint64_t, uint64_t, float, float, bof_fptr, bof_fptr, bof_fptr, char, bof_fptr, bof_fptr, const bof_options *) { bof::set_error("not built yet"); return BOF_EINVAL; }
int bof_flash_csrgemv(char, uint64_t, uint64_t, bof_fptr, bof_fptr, bof_fptr, const float *, float *, const bof_options *) { bof::set_error("not built yet"); return BOF_EINVAL; }
int bof_flash_last_stats(bof_flash_stats *) { return BOF_EINVAL; }
int bof_file_sread(int, uint64_t, uint64_t, uint64_t, uint64_t, void *, int) { return BOF_EINVAL; }
int bof_file_swrite(int, uint64_t, uint64_t, uint64_t, uint64_t, const void *, int) { return BOF_EINVAL; }
}
