// ref_codec_shim.cc -- C exports around the REFERENCE's header-only bit codec.
//
// TEST INFRASTRUCTURE ONLY.  This translation unit includes the reference
// header where it lies (-I$(REFERENCE), fewbit/cpu/codec.h:33-83) and adds
// nothing but extern "C" entry points so tests can call the real
// fewbit::Deflate / fewbit::Inflate through ctypes.  The object it builds
// goes to oracle/_ref/ (git-ignored); no reference source is copied.
#include <cstddef>
#include <cstdint>

#include <fewbit/cpu/codec.h>

extern "C" {

void ref_deflate_u8(const int32_t *codes, size_t n, uint8_t *out, int32_t bitwidth) {
    fewbit::Deflate<uint8_t>(codes, codes + n, out, bitwidth);
}

void ref_inflate_u8(int32_t *codes, size_t n, const uint8_t *in, int32_t bitwidth) {
    fewbit::Inflate<uint8_t>(codes, codes + n, in, bitwidth);
}

}  // extern "C"
