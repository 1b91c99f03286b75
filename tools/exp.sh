tools/runvar.sh pfafter 22
