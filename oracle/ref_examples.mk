# TEST INFRASTRUCTURE: builds the reference's own C callers (rln/ffi_c_examples/*.c, the programs a zerokit user
# compiles against librln) UNMODIFIED, from where they lie under /root/reference, against include/rln.h and
# zerokit_amd/lib/librln.so.  Nothing is copied: the outputs go to oracle/_ref/examples/ (git-ignored; they travel to
# the GPU box with the snapshot, where tests/test_gpu_ffi.py runs them).  Skipped when /root/reference is absent.
REF      ?= /root/reference/rln/ffi_c_examples
OUT      := _ref/examples
NAMES    := basic_proof multi_message_id partial_proof recover_secret stateless type_serialization
CC       ?= gcc
BINS     := $(NAMES:%=$(OUT)/%)

all: $(BINS)

$(OUT)/%: $(REF)/%.c $(REF)/common.c ../include/rln.h ../zerokit_amd/lib/librln.so
	@mkdir -p $(OUT)
	$(CC) -std=c11 -O1 -Wall -Werror -Wno-unused-result -I../include -I$(REF) $< -L../zerokit_amd/lib -lrln \
	    -Wl,-rpath,'$$ORIGIN/../../../zerokit_amd/lib' -o $@

clean:
	rm -rf $(OUT)
.PHONY: all clean
