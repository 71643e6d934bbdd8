"""Instruction-mix / register summary of one kernel from a hipcc -save-temps .s file.
usage: python tools/isa_stats.py <file.s> <kernel-name-substring>"""
import re
import sys

PATS = ["v_cvt_pk_bf16_f32", "v_mfma_", "ds_read_b128", "ds_write_b128", "ds_read_b64 ", "ds_write_b64",
        "global_load_dwordx4", "global_load_dwordx2", "global_load_dword ", "buffer_load", "global_store_dword ",
        "global_store_dwordx4", "s_barrier", "scratch_", "v_sub_f32", "v_pk_add_f32", "v_and_b32", "v_lshlrev_b32",
        "s_waitcnt vmcnt", "s_waitcnt lgkmcnt", "v_accvgpr"]


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2]
    for m in re.finditer(r"^(\w+):\s*; @\1\n(.*?)^\s*\.end_amdhsa_kernel", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if want not in name:
            continue
        print(name)
        for p in PATS:
            n = len(re.findall(re.escape(p), body))
            if n:
                print(f"   {p:26s} {n}")
        for k in (".amdhsa_next_free_vgpr", ".amdhsa_accum_offset", ".amdhsa_group_segment_fixed_size",
                  ".amdhsa_private_segment_fixed_size"):
            mm = re.search(re.escape(k) + r"\s+(\S+)", body)
            print("  ", k, mm.group(1) if mm else None)
        for k in ("num_vgpr", "num_agpr", "private_seg_size"):
            mm = re.search(re.escape(name) + r"\." + k + r",\s*(\S+)", text)
            print("  ", k, mm.group(1) if mm else None)
        mm = re.search(r"; Occupancy:\s*(\d+)", text[m.end():m.end() + 4000])
        print("   occupancy", mm.group(1) if mm else None)


if __name__ == "__main__":
    main()
