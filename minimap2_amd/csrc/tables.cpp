#include <cstdint>
namespace mm2amd {
// ASCII -> nt4: A/a 0, C/c 1, G/g 2, T/t/U/u 3, already-encoded 0..3 map to themselves, anything else 4
// (same mapping as the reference's seq_nt4_table, sketch.c:9-26)
extern const uint8_t kNt4Table[256] = {
#define X16 4,4,4,4,4,4,4,4,4,4,4,4,4,4,4,4
	0,1,2,3, 4,4,4,4, 4,4,4,4, 4,4,4,4, X16, X16, X16,
	4,0,4,1, 4,4,4,2, 4,4,4,4, 4,4,4,4,  4,4,4,4, 3,3,4,4, 4,4,4,4, 4,4,4,4,
	4,0,4,1, 4,4,4,2, 4,4,4,4, 4,4,4,4,  4,4,4,4, 3,3,4,4, 4,4,4,4, 4,4,4,4,
	X16, X16, X16, X16, X16, X16, X16, X16
#undef X16
};
}
