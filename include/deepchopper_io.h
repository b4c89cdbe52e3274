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
 *   dc_pt_open/submit/wait/close  CustomWriter.write_on_batch_end = torch.save   deepchopper/models/callbacks.py:12-25
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
 * stays `readahead` buffers (4 MiB each, default 4 when 0 is passed) ahead of the parser.
 * Environment: DC_IO_INFLATE=zlib selects zlib instead of the in-house DEFLATE decoder; DC_IO_INFLATE_THREADS
 * (default 4, fewer on hosts with < 16 CPUs; 1 = off) decodes one large gzip member on several threads (exactly
 * the serial result: every speculative chunk start is confirmed by the decode that reaches it); DC_IO_INFLATE_CHUNK
 * = compressed bytes per chunk (default 4 MiB; members shorter than four chunks are decoded serially). */
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

/* Same, but `headers` receives the WHOLE definition line after '@' (name, delimiter, description; CR
 * stripped) and name_len[r] the length of its name part.  The chop stage needs both: records that pass
 * through unchanged keep their description (process_chunk returns fq_record.clone(),
 * src/bin/predict.rs:146-164, and the noodles writer emits "@name description"), chopped records are
 * named from the name part alone (Definition::new(id, ""), src/output/split.rs:213-221). */
int dc_fq_read_chunk_full(dc_fq_reader* r, int64_t max_reads, uint8_t* seq, uint8_t* qual, int64_t cap_bases,
                          int64_t* offsets, char* headers, int64_t cap_headers, int64_t* header_offsets,
                          int32_t* name_len, int64_t* n_reads);

/* records delivered so far / compressed-or-plain bytes consumed so far */
int64_t dc_fq_records_read(const dc_fq_reader* r);

/* CPU seconds of the reader's helper threads -- file read-ahead, gzip / BGZF inflate incl. the parallel single-member
 * decoder -- that have FINISHED (exact once the stream is exhausted; a thread still running is not counted yet).  Host-budget
 * accounting of the node's single inflater (bench.py --gpus N, tools/host_budget.py); no reference equivalent. */
double dc_fq_helper_cpu_seconds(const dc_fq_reader* r);

/* Record grammar.  Default (0): what kseq accepts -- `predict` reads with pyfastx.Fastx (deepchopper/data/only_fq.py:34):
 * sequence and quality may wrap over several lines.  strict_four_line = 1: exactly four lines per record, what noodles'
 * fastq reader accepts -- the reference's `chop` reads with it (src/output/writefq.rs:172-199, src/bin/predict.rs:275-316):
 * a record whose quality line is shorter than its sequence is the reference's "Sequence/quality length mismatch" error,
 * never the start of a wrapped quality that swallows the next record.  Set before the first read. */
int dc_fq_set_strict(dc_fq_reader* r, int strict_four_line);
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
/* Marks (new: the reference's chop writes one file from one process, src/bin/predict.rs:266-366; here N per-GPU
 * workers each write a BGZF part and the parent splices them in INPUT order).  dc_bgzf_mark ends the current block
 * and remembers the position; it never waits for the compression workers.  dc_bgzf_close2 = dc_bgzf_close that also
 * hands out the file offset of every mark (ascending; mark k = end of everything written before the k-th call) and
 * the number of data bytes in front of the 28-byte EOF block (and the CPU seconds the compression workers spent in
 * deflate: the host-budget accounting of an 8-rank run).  A part cut at its marks is a series of whole BGZF blocks, so
 * any concatenation of such pieces + one EOF block is a valid BGZF file. */
int dc_bgzf_mark(dc_bgzf_writer* w);
int64_t dc_bgzf_mark_count(const dc_bgzf_writer* w);
int dc_bgzf_close2(dc_bgzf_writer* w, int64_t* mark_offsets, int64_t cap, int64_t* n_marks, int64_t* data_bytes,
                   double* deflate_cpu_seconds);

/* Raw DEFLATE (RFC 1951) of one block of fewer than 65 535 bytes by this library's own encoder (csrc/fast_deflate.h) -- what
 * dc_bgzf_* puts inside every BGZF block at the default level 6 (noodles / flate2's default in the reference's writer,
 * src/output/writefq.rs:252-276): zlib level 6's ratio on nanopore FASTQ at 2.5 times its speed (two match finders, matches
 * taken only where they are cheaper than their literals, one dynamic-Huffman block; a stored block if that is smaller).
 * Deterministic: the same bytes for the same input, whatever ran before.  Returns DC_FQ_ERR_ARG if `cap` is too small
 * (n + 5 always suffices). */
int dc_deflate_block(const uint8_t* in, int64_t n, uint8_t* out, int64_t cap, int64_t* out_len);

/* ---- chop stage of the fused predict -> chop flow ------------------------------------------------
 * One call per reader chunk: the gating of process_chunk (src/bin/predict.rs:130-192), the record
 * splitting of split_noodle_records_by_remove_intervals (src/output/split.rs:171-226) or, with
 * output_chopped_seqs, split_noodle_records_by_intervals (src/output/split.rs:138-169), and the FASTQ
 * text of every resulting record written to `w`.  Replaces the reference's `.pt` round trip
 * (callbacks.py:12-25 -> load_predicts_from_batch_pts, src/bin/predict.rs:222-235) for the case where
 * predict and chop run in one process: the per-read inputs are what dc_smooth_intervals_from_logits
 * (deepchopper_hip.h) leaves on the device, copied to the host by the caller. */
typedef struct dc_chop_params {
  int32_t min_read_len;               /* default::MIN_READ_LEN = 150 (src/default.rs)                 */
  int32_t max_process_intervals;      /* --mpi, 4                                                      */
  int32_t min_read_length_after_chop; /* --mcr, 20                                                     */
  int32_t chop_type;                  /* --ct: 0 terminal, 1 internal, 2 all                           */
  int32_t output_chopped_seqs;        /* --ocq: emit the chopped segments instead of the kept ones     */
  int32_t id_annotation;              /* "|T" / "|I" suffix (process_chunk passes true)                */
} dc_chop_params;

/*   headers / header_offsets / name_len   as dc_fq_read_chunk_full delivers them (name_len NULL: every
 *                                         header is a bare name)
 *   seq / qual / offsets                  the chunk's packed reads (raw FASTQ bytes)
 *   pred_len     int32[n]                 length of the prediction (= tokens - 1); a read whose prediction
 *                                         is shorter than the read (truncated) passes through unchanged
 *   n_intervals  int32[n], intervals int32[n][max_intervals][2]   smoothed, selected intervals (0 = none /
 *                                         more than the approved number)
 *   n_keep       int32[n], keep int32[n][max_keep][2]             complement segments (generate_unmaped_intervals)
 *   records_per_read  int32[n] or NULL    records written for each read
 * Chopped records take their bases from the prediction's sequence as the reference does (Predict.seq =
 * id_list2seq(tokens): upper-case ACGT, U -> T, anything else N); pass-through records are the input
 * record byte for byte. */
int dc_chop_write_chunk(dc_bgzf_writer* w, const dc_chop_params* p, int64_t n, const char* headers,
                        const int64_t* header_offsets, const int32_t* name_len, const uint8_t* seq,
                        const uint8_t* qual, const int64_t* offsets, const int32_t* pred_len,
                        const int32_t* n_intervals, const int32_t* intervals, int32_t max_intervals,
                        const int32_t* n_keep, const int32_t* keep, int32_t max_keep,
                        int32_t* records_per_read, int64_t* n_records_out);

/* ---- the predict -> chop wire format ---------------------------------------------------------------
 * CustomWriter.write_on_batch_end (deepchopper/models/callbacks.py:12-25) = torch.save of the predict step's dict
 * {prediction f32[B,L,2], target i64[B,L], seq i64[B,L], qual f32[B,L], id i64[B,256]} as `{dir}/{rank}_{index}.pt`, one
 * file per reference batch; the consumers are `deepchopper chop` / load_predicts_from_batch_pt
 * (src/smooth/predict.rs:264-317: candle's pickle::read_all) and torch.load.  dc_pt_* writes the same container (ZIP of
 * STORED entries, the pickle stream torch.save emits for this dict, storages 64-byte aligned) natively on a pool of
 * threads: torch.save's byte-wise CRC-32 and per-file Python made the wire format, not the GPU, the bound of
 * `deepchopper predict` (754 MB/s against the 2.2 GB/s the GPU path produces on a nanopore-like length mix).
 *
 * dc_pt_submit queues the files of ONE fused launch and returns at once:
 *   row0 / n_rows / col0  int32[n_files]   file i holds rows [row0, row0 + n_rows) of the launch from column col0 on
 *                                           (the column its own left padding starts at): L = row_stride - col0
 *   logits f32[n_total_rows][row_stride][2], labels i8[..][row_stride] (-100 = ignore), ids u8[..][row_stride] (token ids),
 *   quals f32[..][row_stride]               the launch's dense HOST arrays -- borrowed until dc_pt_wait(ticket) returns
 *   names / name_offsets[n_total_rows + 1], truncated u8[n_total_rows]   read ids -> the id rows [len, truncated, ord(c)...]
 *                                           (tokenizer.py:170-175); an id longer than 127 bytes or a byte above 127 fails
 *                                           as the reference's int8 cast does (tokenizer.py:91), DC_FQ_ERR_FORMAT
 * dc_pt_wait(ticket) returns once every file of that submit and of all earlier ones is written (first error wins, message in
 * dc_io_last_error); dc_pt_close waits for everything and frees the handle.  One handle per caller thread. */
typedef struct dc_pt_writer dc_pt_writer;
int dc_pt_open(int threads, dc_pt_writer** out);
int dc_pt_submit(dc_pt_writer* w, const char* dir, int32_t rank, int64_t first_index, int32_t n_files, const int32_t* row0,
                 const int32_t* n_rows, const int32_t* col0, const float* logits, const int8_t* labels, const uint8_t* ids,
                 const float* quals, int64_t n_total_rows, int64_t row_stride, const char* names,
                 const int64_t* name_offsets, const uint8_t* truncated, int64_t* ticket_out);
/* The same files from payloads the GPU packed (include/deepchopper_hip.h: dc_pt_pack, copied to host memory by the caller):
 * file i = rows [row0[i], +n_rows[i]) of the launch (for its id rows), n_cols[i] columns, its storages data/0..3 back to
 * back at payload + payload_offset[i] (28 B per position).  Nothing is copied or converted on the host: a file goes out
 * as one writev() of [headers | storage | header | storage ...], the CRC-32 is the only pass over the data.  `payload` is
 * borrowed until dc_pt_wait(ticket) returns. */
int dc_pt_submit_packed(dc_pt_writer* w, const char* dir, int32_t rank, int64_t first_index, int32_t n_files,
                        const int32_t* row0, const int32_t* n_rows, const int32_t* n_cols, const int64_t* payload_offset,
                        const void* payload, int64_t payload_bytes, int64_t n_total_rows, const char* names,
                        const int64_t* name_offsets, const uint8_t* truncated, int64_t* ticket_out);
int dc_pt_wait(dc_pt_writer* w, int64_t ticket);
int dc_pt_close(dc_pt_writer* w);

/* The consumer side of the wire format for the chop driver: load_predicts_from_batch_pts / _pt
 * (src/smooth/predict.rs:211-317) over a list of `.pt` files on `threads` threads -- per read its id (the id row's
 * [len, truncated, ord(c)...]), truncation flag and the argmax labels (first maximum wins) over the positions whose target is
 * not `ignore_label`.  Files are read through mmap by their ZIP central directory and storage keys (`<stem>/data/0` prediction
 * f32[B][L][2], `<stem>/data/1` target i64[B][L], `<stem>/data/4` id i64[B][256]; B and L follow from the storage sizes).  A file that
 * is not laid out that way (or cannot be opened) is counted in n_failed and contributes no reads: the caller hands it to
 * its general path (torch.load), which also reports the reference's "load pt ... fail" line.
 * dc_pt_export copies the set into caller buffers sized from dc_pt_load's totals, reads in file order:
 *   labels i8[label_bytes], label_offsets i64[n_reads + 1], names char[name_bytes], name_offsets i64[n_reads + 1],
 *   truncated u8[n_reads], file_of_read i32[n_reads], file_failed u8[n_paths]      (any of the last three may be NULL). */
typedef struct dc_pt_set dc_pt_set;
int dc_pt_load(const char* const* paths, int64_t n_paths, int32_t threads, int64_t ignore_label, dc_pt_set** out,
               int64_t* n_reads, int64_t* label_bytes, int64_t* name_bytes, int64_t* n_failed);
int dc_pt_export(const dc_pt_set* set, int8_t* labels, int64_t* label_offsets, char* names, int64_t* name_offsets,
                 uint8_t* truncated, int32_t* file_of_read, uint8_t* file_failed);
void dc_pt_free(dc_pt_set* set);

#ifdef __cplusplus
}
#endif
#endif /* DEEPCHOPPER_IO_H */
