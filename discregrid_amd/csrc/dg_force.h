// dg_force.h -- DG_FORCE: the ONE environment variable behind which every test hook and tuning knob of the library lives.
//
//     DG_FORCE="key=value;key=value;..."      (separators ';' or blanks; a value may contain commas)
//
// The documented environment of the library is the short table in INTEGRATION.md (DG_DEVICES, DG_FORCE_CPU, ...); what a
// parity test needs to steer a launch down a particular path (k1_fast=0: the exact K1 kernel; heavy_work=4: every brick
// through the heavy-brick kernels; k2_rows=0; k3_cells=0; ...) and what a measuring script needs to vary (k3_rb1=6; xcd_chunk=512;
// field_fractions=0.2,0.3,0.5) are keys of this variable, read afresh on every call (a launch costs microseconds, the parse
// nanoseconds).  Nothing selects a different ALGORITHMIC result: every path is bit-identical by construction and by test.
// Header-only: the kernel library and the C++ host library both include it.
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <string>

namespace dg
{
// value of `key` in DG_FORCE, or false
inline bool force_lookup(const char* key, std::string& value)
{
	const char* e = std::getenv("DG_FORCE");
	if (!e)
		return false;
	const size_t klen = std::strlen(key);
	for (const char* p = e; *p;)
	{
		while (*p == ';' || *p == ' ' || *p == '\t')
			++p;
		const char* end = p;
		while (*end && *end != ';' && *end != ' ' && *end != '\t')
			++end;
		if ((size_t)(end - p) > klen && std::strncmp(p, key, klen) == 0 && p[klen] == '=')
		{
			value.assign(p + klen + 1, end);
			return true;
		}
		p = end;
	}
	return false;
}
inline bool force_set(const char* key)
{
	std::string v;
	return force_lookup(key, v);
}
inline int force_int(const char* key, int fallback, int lo, int hi)
{
	std::string v;
	if (!force_lookup(key, v))
		return fallback;
	return std::max(lo, std::min(hi, std::atoi(v.c_str())));
}
} // namespace dg
