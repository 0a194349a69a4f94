// bf16-storage engine (the production path)
#include "engine_impl.h"

namespace ach {
EngineBase* make_engine_bf16(const ach_config& cfg) { return new Engine<bf16_t>(cfg); }
}  // namespace ach
