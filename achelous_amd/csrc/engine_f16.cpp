// fp16-storage engine (round 4): the production path.  Same bytes and MFMA rate as bf16 storage, 11 mantissa bits instead of 8 — and the
// type the reference's own mixed-precision mode computes in (utils/utils_fit.py:120-121).  Inputs / outputs fp16, or bf16 with option "io_bf16".
#include "engine_impl.h"
namespace ach {
EngineBase* make_engine_f16(const ach_config& cfg) { return new Engine<f16_t>(cfg); }
}  // namespace ach
