// Raw little-endian (de)serialisation of standard-layout values -- same wire format as the
// reference's utility/serialize.hpp:11-37 (bytes of the object, no header, no padding added).
#pragma once
#include <cstddef>
#include <streambuf>
#include <type_traits>

namespace Discregrid
{
namespace serialize
{
template <class T>
bool write(std::streambuf& buf, T const& val)
{
	static_assert(std::is_standard_layout<T>::value, "data is not standard layout");
	return (std::size_t)buf.sputn(reinterpret_cast<const char*>(&val), sizeof(T)) == sizeof(T);
}
template <class T>
bool read(std::streambuf& buf, T& val)
{
	static_assert(std::is_standard_layout<T>::value, "data is not standard layout");
	return (std::size_t)buf.sgetn(reinterpret_cast<char*>(&val), sizeof(T)) == sizeof(T);
}
} // namespace serialize
} // namespace Discregrid
