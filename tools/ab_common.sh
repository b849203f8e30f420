# sourced by the ab_*.sh scripts: A/B of library variants on ONE GPU box.  The variants live under ab_libs/ at the repo root
# (git-ignored, NOT gpurun-ignored: put them there for the one call that needs them, remove them afterwards -- every file under the
# repo is pushed to the box with every lease).  The installed library is saved to a private temporary file and put back by a trap on
# EXIT, whatever ends the script (set -e, a timeout, a kill); every line names the sha256 of the library it was measured on.
AB_DIR=${AB_DIR:-ab_libs}
AB_LIB=sqeazy_amd/lib/libsqeazy_amd.so
AB_SAVED=$(mktemp /tmp/sqy_installed.XXXXXX.so)
cp "$AB_LIB" "$AB_SAVED"
trap 'cp "$AB_SAVED" "$AB_LIB"; rm -f "$AB_SAVED"' EXIT
ab_install() {      # ab_install <tag>: puts ab_libs/<tag>.so in place, prints "<tag> <sha256 prefix>"
    cp "$AB_DIR/$1.so" "$AB_LIB"
    AB_SHA=$(sha256sum "$AB_LIB" | cut -c1-12)
}
mkdir -p gpurun_out
