// rover_internal.hpp -- symbols shared between the translation units of librover_hip.so (not part of the C ABI).
#pragma once

// records the text rover_last_error() returns on this thread and hands `code` back
__attribute__((visibility("hidden"))) int rover_internal_fail(int code, const char *fmt, const char *detail = "");
