// falcon_gpu.hpp -- the MI355X verifier beside the reference in one translation unit.
//
// falcon_amd.hpp also exports its functions under the reference's own namespace `falcon` (falcon.hpp:230-249) so that
// it can replace the reference's header; a program that includes the reference's falcon.hpp as well would then see
// two falcon::verify<N>. This header keeps everything in namespace falcon_gpu only:
//
//   #include "falcon.hpp"       // the reference (CPU): falcon::keygen, falcon::sign, falcon::verify
//   #include "falcon_gpu.hpp"   // this library (GPU):  falcon_gpu::verify, falcon_gpu::verify_many, ...
//
// A translation unit includes either this header or falcon_amd.hpp first; whichever comes first decides.
#pragma once
#ifndef FALCON_AMD_NO_REFERENCE_NAMESPACE
#define FALCON_AMD_NO_REFERENCE_NAMESPACE 1
#endif
#include "falcon_amd.hpp"
