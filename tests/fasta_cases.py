"""Hand-built FASTA inputs shared by the CPU tests, the GPU tests and tests/golden/make_golden.py (section 5).
Data only: every case is a file image; what the reference makes of it downstream of the parser is in
tests/golden/fasta_cases.npz."""

CASES = {
    "plain": b">a one\nACGT\nAC\n>b\nGGGG\n",
    "crlf": b">a one \r\nACGT\r\nAC\r\n>b\t tab title \r\nGG GG\r\n",
    "no_final_newline": b">a\nACGT\n>b\nTT",
    "header_at_eof": b">a\nACGT\n>b",
    "header_at_eof_nl": b">a\nACGT\n>b\n",
    "empty_records": b">a\n>b\n\n>c\nAC\n\n\n>d\n",
    "leading_blank": b"\n  \n\r\n>a\nAC\n",
    "inner_gt": b">a\nAC>GT\nA>\n>b\nTT\n",
    "spaces_inside": b">a\nA C G T\n  ACGT  \n",
    "lowercase_n": b">a\nacgtnNNNNacgt\n",
    "only_blank": b"\n\n  \n",
    "single_byte_lines": b">a\nA\nC\nG\nT\n",
    # IUPAC ambiguity codes, RNA 'U' in both cases, gaps, digits and lower case inside wrapped records
    "iupac_u_lowercase": (b">iupac every ambiguity code\nACGTRYKMSWBDHVNacgtrykmswbdhvnACGTTGCA\nGATTACAGATTACA\n"
                          b">rna\nACGUACGUUUGCAUGCAacguacguACGTTGCAugca\nUUUUACGTACGTACGTUUUU\n"
                          b">mixed case  \r\nacgtACGTacgtAcGtaCgTACGTACGTTTGACCA\r\nnnACGTACGTnn\r\n"
                          b">gaps and digits\nACGT-ACGT*ACGT.ACGT1234ACGTACGTAC\n--ACGTTGCAAC--\n"
                          b">short\nAC\n>empty\n>long tail\n" + b"ACGTTGCAGGATCCAT" * 40 + b"\n"),
}

# patterns x strands the reference's compute_frequency is run on for every parsed record (make_golden.py section 5)
PROFILE_KEYS = [("1111", "both"), ("1101", "minus"), ("111", "plus"), ("11011", "both")]
