// fp32-storage engine (the exact-parity path)
#include "engine_impl.h"

namespace ach {
EngineBase* make_engine_f32(const ach_config& cfg) { return new Engine<float>(cfg); }
}  // namespace ach
