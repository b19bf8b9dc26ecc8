/* oracle/ref_stubs/png++/image.hpp -- checker build only.  A stand-in for png++'s image<gray_pixel_16> that lets the reference's
 * readPNG16 / writePNG16 (adcensus.cu:1670-1704) RUN: what those functions define is the pixel arithmetic (val == 0 ? 0 : val / 256;
 * (uint16_t)(val < 1e-5 ? 0 : val * 256)) -- the PNG container is png++ / libpng's, third-party code absent from /root/reference.
 * So the stand-in keeps the pixels and exchanges them with the tests in a trivial container instead of a PNG:
 *   "MCREF16 <width> <height>\n" followed by width * height little-endian uint16 pixels, rows top to bottom.
 * tests/test_hostio_ref.py converts between that and real PNG files (through libpng, PIL) to compare with mc_read_png16 / mc_write_png16. */
#ifndef MCREF_PNGPP_H
#define MCREF_PNGPP_H
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
namespace png {
typedef uint16_t gray_pixel_16;
template <typename P> class image {
	int w_, h_;
	std::vector<P> px_;
public:
	explicit image(const char *fname) : w_(0), h_(0)
	{
		FILE *f = fopen(fname, "rb");
		if (!f || fscanf(f, "MCREF16 %d %d", &w_, &h_) != 2 || fgetc(f) != '\n' || w_ < 1 || h_ < 1) abort();
		px_.resize((size_t)w_ * h_);
		if (fread(px_.data(), sizeof(P), px_.size(), f) != px_.size()) abort();
		fclose(f);
	}
	image(int w, int h) : w_(w), h_(h), px_((size_t)w * h) {}
	int get_width() const { return w_; }
	int get_height() const { return h_; }
	P get_pixel(int x, int y) const { return px_[(size_t)y * w_ + x]; }
	void set_pixel(int x, int y, P v) { px_[(size_t)y * w_ + x] = v; }
	void write(const char *fname)
	{
		FILE *f = fopen(fname, "wb");
		if (!f) abort();
		fprintf(f, "MCREF16 %d %d\n", w_, h_);
		fwrite(px_.data(), sizeof(P), px_.size(), f);
		fclose(f);
	}
};
}  // namespace png
#endif
