// placeholder main; replaced by the real harness
int main() { return 0; }
