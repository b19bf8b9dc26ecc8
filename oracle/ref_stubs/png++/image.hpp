/* oracle/ref_stubs/png++/image.hpp -- checker build only.  readPNG16 / writePNG16 (adcensus.cu:1670-1705)
 * are not on the predict path; this inert stand-in only lets the file compile.  Using it aborts. */
#ifndef MCREF_PNGPP_H
#define MCREF_PNGPP_H
#include <stdint.h>
#include <stdlib.h>
namespace png {
typedef uint16_t gray_pixel_16;
template <typename P> class image {
public:
	explicit image(const char *) { abort(); }
	image(int, int) { abort(); }
	int get_width() const { return 0; }
	int get_height() const { return 0; }
	P get_pixel(int, int) const { return P(); }
	void set_pixel(int, int, P) {}
	void write(const char *) {}
};
}  // namespace png
#endif
