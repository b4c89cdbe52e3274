/* deepchopper_io.h -- host-side FASTQ / gzip / BGZF streaming I/O for the predict -> chop path
 * ("the stage either side of the path; bounds 8-GPU throughput").
 *
 * Plain C ABI, no GPU needed, built as libdeepchopper_io.so from deepchopper_amd/csrc/fastq_io.cpp
 * (g++ + zlib + pthreads).  What each entry point replaces in the reference:
 *
 *   dc_io_check_compressed_type   check_compressed_type            src/output/writefq.rs:84-147
 *   dc_fq_open / dc_fq_read_chunk StreamingFastqReader::new/next    src/output/writefq.rs:172-199
 *                                 (+ create_reader_for_compressed_file :148-161: plain | gzip | BGZF,
 *                                 anything else -> "unsupported compression type"), and the record
 *                                 validation of parse_fastq_file     deepchopper/data/only_fq.py:21-85
 *   dc_fq_open_range              (no reference equivalent) byte-range shard of an uncompressed FASTQ
 *                                 so that each GPU rank parses only its own contiguous slice
 *   dc_bgzf_open/write/close      noodles bgzf::io::MultithreadedWriter as used by
 *                                 write_fq_parallel_for_noodle_record  src/output/writefq.rs:252-276
 *                                 and the chop driver                  src/bin/predict.rs:266-270
 *   dc_bgzf_write_fastq           fastq::io::Writer::write_record ("@name\nseq\n+\nqual\n") over it
 *
 * Conventions: every function returns 0 on success, non-zero on failure with a thread-local message in
 * dc_io_last_error().  Buffers are caller-owned; the only allocations the caller must release are the
 * opaque handles (dc_fq_close, dc_bgzf_close).  Handles are not thread-safe; different handles may be
 * used from different threads concurrently.
 */
#ifndef DEEPCHOPPER_IO_H
#define DEEPCHOPPER_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* dc_io_last_error(void);

/* CompressedType (src/output/writefq.rs:51-60), same order */
enum dc_compressed_type {
  DC_CT_UNCOMPRESS = 0,
  DC_CT_GZIP = 1,
  DC_CT_BGZIP = 2,
  DC_CT_ZIP = 3,
  DC_CT_BZIP2 = 4,
  DC_CT_XZ = 5,
  DC_CT_ZSTD = 6,
  DC_CT_UNKNOWN = 7
};
int dc_io_check_compressed_type(const char* path, int* type_out);

/* ---- streaming FASTQ reader ------------------------------------------------------------------- */
typedef struct dc_fq_reader dc_fq_reader;

/* status codes of the reader (also the return value of dc_fq_read_chunk on failure) */
enum dc_fq_status {
  DC_FQ_OK = 0,
  DC_FQ_ERR_IO = 1,          /* open/read/inflate failure, unsupported compression            */
  DC_FQ_ERR_FORMAT = 2,      /* record does not start with '@' / no '+' line / truncated record */
  DC_FQ_ERR_INCOMPLETE = 3,  /* empty name, sequence or quality   (only_fq.py:38-40)            */
  DC_FQ_ERR_LENGTH = 4,      /* len(seq) != len(qual)             (only_fq.py:43-45)            */
  DC_FQ_ERR_ARG = 5
};

/* Opens plain / gzip (incl. multi-member) / BGZF FASTQ.  Decompression runs on a helper thread that
 * stays `readahead` buffers (4 MiB each, default 4 when 0 is passed) ahead of the parser. */
int dc_fq_open(const char* path, int readahead, dc_fq_reader** out);

/* Byte-range shard of an UNCOMPRESSED file: yields exactly the records whose '@' line starts in
 * [byte_begin, byte_end).  The reader synchronises to the first record boundary at or after
 * byte_begin (a line starting with '@' whose next-but-one line starts with '+'), so ranges that
 * tile [0, file size) yield every record exactly once.  Compressed input is rejected (DC_FQ_ERR_ARG). */
int dc_fq_open_range(const char* path, int64_t byte_begin, int64_t byte_end, int readahead, dc_fq_reader** out);

/* Reads up to max_reads records, stopping early before a record that would not fit cap_bases /
 * cap_names (a single record larger than a capacity is an error).  Output, all caller-owned:
 *   seq, qual      uint8[cap_bases]   concatenated sequence / quality bytes (no separators; CR stripped)
 *   offsets        int64[max_reads+1] read r occupies [offsets[r], offsets[r+1])
 *   names          char[cap_names]    concatenated names (definition line after '@' up to the first blank or tab)
 *   name_offsets   int64[max_reads+1]
 *   *n_reads       records delivered; 0 = end of input
 * This is exactly the (seq, qual, offsets) triple dc_tokenize_batch takes (deepchopper_hip.h). */
int dc_fq_read_chunk(dc_fq_reader* r, int64_t max_reads, uint8_t* seq, uint8_t* qual, int64_t cap_bases,
                     int64_t* offsets, char* names, int64_t cap_names, int64_t* name_offsets, int64_t* n_reads);

/* records delivered so far / compressed-or-plain bytes consumed so far */
int64_t dc_fq_records_read(const dc_fq_reader* r);
int64_t dc_fq_bytes_consumed(const dc_fq_reader* r);
void dc_fq_close(dc_fq_reader* r);

/* ---- multithreaded BGZF writer ---------------------------------------------------------------- */
typedef struct dc_bgzf_writer dc_bgzf_writer;

/* threads >= 1 compression workers; level 1..9 (0 -> 6, zlib's default as noodles uses).
 * The file is a valid multi-member gzip: any gzip reader, and the reference's
 * bgzf::io::Reader / StreamingFastqReader, read it back. */
int dc_bgzf_open(const char* path, int threads, int level, dc_bgzf_writer** out);
int dc_bgzf_write(dc_bgzf_writer* w, const void* data, int64_t n);
/* n records from packed buffers, each written as "@name\nseq\n+\nqual\n" (noodles fastq writer) */
int dc_bgzf_write_fastq(dc_bgzf_writer* w, int64_t n, const char* names, const int64_t* name_offsets,
                        const uint8_t* seq, const uint8_t* qual, const int64_t* offsets);
/* flushes, appends the 28-byte BGZF EOF block, closes the file and frees the handle */
int dc_bgzf_close(dc_bgzf_writer* w);

#ifdef __cplusplus
}
#endif
#endif /* DEEPCHOPPER_IO_H */
