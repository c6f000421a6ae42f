/*
 * meryl_seq.h -- C ABI of the sequence-file loader that feeds the count path.
 *
 * Replaces, for FASTA/FASTQ/SAM (plain, gzip or BGZF) and BAM, the reference's
 *   openSequenceFile(name)                     src/meryl/merylOp.C:200
 *   dnaSeqFile::loadBases(seq, maxLength, seqLength, endOfSequence)
 *                                              src/meryl/merylInput.C:257
 * (both live in the absent submodule marbl/meryl-utility, utility/src/sequence/).
 * The contract is the one merylInput::loadBases documents
 * (src/meryl/merylInput.H:67-70): bases only -- no headers, no qualities --
 * of ONE sequence per call, at most max_length of them; *end_of_sequence
 * tells whether the sequence ended inside this call; returns 0 at end of file.
 * SAM and BAM (vendored htslib in the reference, src/main.mk:92-140; README.md:11
 * "Direct kmer counting from bam / cram") give one sequence per alignment record:
 * SEQ as stored, no record filtered by its flags -- which records the reference
 * keeps is decided inside the absent submodule, so that choice is unpinned.
 * .bz2 and .xz go through `bzip2 -dc` / `xz -dc` pipes.  CRAM (needs the reference
 * genome and htslib's codecs) goes through a `samtools view -h` pipe and the SAM path
 * when that binary is on the PATH; without it msr_open refuses the file with a message.
 */
#ifndef MERYL_SEQ_H
#define MERYL_SEQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msr_reader msr_reader;

#define MSR_FORMAT_FASTX 0   /* FASTA / FASTQ text (msr_read_text may hand it to the device parser) */
#define MSR_FORMAT_BAM   1
#define MSR_FORMAT_SAM   2

/* name "-" reads stdin.  NULL on failure; text via msr_last_error().  A BGZF file (bgzip output, every BAM) is
 * inflated block-parallel by up to 16 threads (MERYL_BGZF_THREADS=n; 0 = zlib's single stream). */
msr_reader *msr_open(const char *name);
void        msr_close(msr_reader *r);
const char *msr_last_error(void);

/* 1 = got something (possibly 0 bases with *end_of_sequence set for an empty
 * sequence), 0 = end of input, <0 = malformed input. */
int msr_load_bases(msr_reader *r, char *seq, uint64_t max_length, uint64_t *seq_length, int *end_of_sequence);

/* Many sequences per call: the bases of consecutive sequences with a '.' after each one that ended -- the stream
 * merylOp-countThreads.C:138-231 assembles from loadBases calls ('.' breaks k-mers), ready for mgc_push_bases with
 * end_of_sequence = 0.  Fills up to max_length bytes; 1 = got something, 0 = end of input, <0 = malformed input.
 * Short reads cost one C call per 2 MiB instead of one per read. */
int msr_load_stream(msr_reader *r, char *buf, uint64_t max_length, uint64_t *length);

/* The file's TEXT (decompressed, otherwise untouched), up to max_length bytes per call; 0 at end of input, <0 on a
 * read error.  For callers that parse on the device (mgc_push_text, include/meryl_gpu_count.h); not to be mixed with
 * msr_load_bases on the same reader. */
int64_t msr_read_text(msr_reader *r, char *buf, uint64_t max_length);

/* MSR_FORMAT_*: told from the content (BAM magic, @HD line) or the name (.sam). */
int msr_format(const msr_reader *r);

/* 1 when the file name ends in .gz or .bam (the reference reserves a second loader
 * thread for it, src/meryl/merylOp-countThreads.C:162-168). */
int msr_is_compressed(const msr_reader *r);

/* guesstimateNumberOfkmersInInput_dnaSeqFile, src/meryl/merylOp-count.C:410-433:
 * file size x1 (plain) x3 (.gz) x3.5 (.bz2) x4 (.xz); 0 for "-". */
uint64_t msr_guess_number_of_kmers(const char *name);

#ifdef __cplusplus
}
#endif
#endif
