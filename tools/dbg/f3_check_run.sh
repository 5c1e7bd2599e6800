for tag in "$@"; do echo "== $tag"; MVOC_HIP_LIB=$PWD/tools/lab/libmvoc_attn_$tag.so timeout 300 python tools/dbg/f3_check.py 2>&1 | grep -v amdgpu | tr '\n' ';'; echo; done
