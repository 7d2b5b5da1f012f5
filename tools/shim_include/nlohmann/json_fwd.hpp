// Forward declaration only — what <nlohmann/json_fwd.hpp> provides and all that src/util/config.h:20 needs from it.
// nlohmann/json is a submodule the reference mount does not carry; this is not a stand-in for the library.
#pragma once
namespace nlohmann { class json; }
